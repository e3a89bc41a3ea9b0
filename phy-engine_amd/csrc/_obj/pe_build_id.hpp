#define PE_BUILD_ID "6afc8e02b2fce425"
