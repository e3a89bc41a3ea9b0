#define PE_BUILD_ID "947030cfcd751b7d"
