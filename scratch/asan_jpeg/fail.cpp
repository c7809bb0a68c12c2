#include <cstdarg>
struct icl_ctx;
int icl_fail(icl_ctx *, int code, const char *, ...) { return code; }
