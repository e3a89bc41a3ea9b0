#define PE_BUILD_ID "a8d1181b87c4ad9a"
