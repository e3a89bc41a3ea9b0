#define PE_BUILD_ID "0d989e1dedfa23bf"
