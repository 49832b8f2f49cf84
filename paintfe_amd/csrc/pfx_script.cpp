// pfx_script.cpp — placeholder until the Rhai call-statement front-end lands (B5/B6).
#include "pfx_internal.h"
extern "C" {
int pfx_script_run(pfx_ctx* ctx, const char*, uint8_t*, uint32_t, uint32_t, const uint8_t*, pfx_script_result* r)
{
    if (r) std::memset(r, 0, sizeof *r);
    return pfx_fail(ctx, PFX_ERR_UNSUPPORTED, "script front-end not built yet");
}
int pfx_cli_main(int, char**) { return 2; }
}
