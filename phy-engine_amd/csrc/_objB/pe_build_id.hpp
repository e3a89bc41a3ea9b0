#define PE_BUILD_ID "3ef9bba03b264cb5"
