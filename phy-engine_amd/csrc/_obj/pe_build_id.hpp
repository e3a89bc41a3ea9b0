#define PE_BUILD_ID "ef6dec4640c1f2f4"
