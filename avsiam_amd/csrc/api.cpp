// C-ABI housekeeping of libavsiam_hip.so: error string, version, tuning knobs.  Every entry point returns 0 on success,
// a negative code on failure (-1 launch/runtime error, -2 bad argument) and never throws, allocates or
// synchronises; avs_last_error() describes the most recent failure on the calling thread.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void avs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* avs_last_error(void) { return g_err; }

// 2 (round 6): avs_attn_fwd / avs_attn_bwd need 16-byte-aligned rows (ldo % 8 == 0; was % 4), avs_gemm_nt_fp8 gives out_f32 == 2 / a_e5m2 == 2
// a meaning (gelu'(x) as 8-bit codes), "ln_dma" is 0 | 1, the deterministic-reduction knob "det" exists
extern "C" int avs_abi_version(void) { return 2; }

// number of compute units of the current device (used by hosts to size split factors); <0 on error
extern "C" int avs_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

// ---- tuning knobs (common.h AvsTuning): the library's only global state besides the thread-local error string.  Written by
// avs_tuning_set (the host binding calls it once per knob at load, from the AVSIAM_* environment) before kernels are queued; launchers
// read them, nothing is initialised lazily and nothing reads the environment.
static AvsTuning g_tuning = {/*gemm_tile*/ 0, /*gemm_persistent*/ 1, /*gemm_nt8*/ 1, /*nt_tile_h*/ 0, /*nt_grid*/ 0, /*cu_reserve*/ 0,
                             /*ln_dma*/ 1, /*ln_rpw*/ 0, /*gemm_ring*/ 2, /*attn_ring*/ 0, /*nt_big_min*/ 0, /*det*/ 0};      // (attn_ring: measured neutral to slower in round 5 - DESIGN.md 5e - so off by default)
AvsTuning& avs_tuning() { return g_tuning; }

int avs_persistent_slots() {
    static int ncu = 0;                 // a device property, not a knob
    if (ncu == 0) {
        const int n = avs_device_cu_count();
        ncu = n > 0 ? n : 256;
    }
    const int s = ncu - g_tuning.cu_reserve;
    return s < 8 ? 8 : s;
}

struct Knob { const char* name; int AvsTuning::*field; int lo, hi; };
static const Knob g_knobs[] = {
    {"gemm_tile", &AvsTuning::gemm_tile, 0, 256},   {"gemm_persistent", &AvsTuning::gemm_persistent, 0, 1}, {"gemm_nt8", &AvsTuning::gemm_nt8, 0, 1},
    {"nt_tile_h", &AvsTuning::nt_tile_h, 0, 256},   {"nt_grid", &AvsTuning::nt_grid, 0, 1 << 20},           {"cu_reserve", &AvsTuning::cu_reserve, 0, 128},
    {"ln_dma", &AvsTuning::ln_dma, 0, 1},           {"ln_rpw", &AvsTuning::ln_rpw, 0, 16},                  {"attn_ring", &AvsTuning::attn_ring, 0, 1},
    {"gemm_ring", &AvsTuning::gemm_ring, 0, 2},     {"nt_big_min", &AvsTuning::nt_big_min, 0, 1 << 20},     {"det", &AvsTuning::det, 0, 1},
};

extern "C" int avs_tuning_set(const char* name, int value) {
    AVS_CHECK_ARG(name, "tuning_set: null name");
    for (const Knob& k : g_knobs)
        if (!strcmp(k.name, name)) {
            AVS_CHECK_ARG(value >= k.lo && value <= k.hi, "tuning_set: %s = %d outside [%d, %d]", name, value, k.lo, k.hi);
            AVS_CHECK_ARG(strcmp(name, "gemm_tile") || value == 0 || value == 128 || value == 256, "tuning_set: gemm_tile must be 0 (auto), 128 or 256");
            AVS_CHECK_ARG(strcmp(name, "nt_tile_h") || value == 0 || value == 256 || value == 224 || value == 240, "tuning_set: nt_tile_h must be 0 (auto), 256, 224 or 240");
            AVS_CHECK_ARG(strcmp(name, "ln_rpw") || value == 0 || value == 4 || value == 8 || value == 16, "tuning_set: ln_rpw must be 0 (auto), 4, 8 or 16");
            g_tuning.*(k.field) = value;
            return 0;
        }
    avs_set_error("tuning_set: unknown knob '%s'", name);
    return -2;
}

extern "C" int avs_tuning_get(const char* name, int* value) {
    AVS_CHECK_ARG(name && value, "tuning_get: null argument");
    for (const Knob& k : g_knobs)
        if (!strcmp(k.name, name)) {
            *value = g_tuning.*(k.field);
            return 0;
        }
    avs_set_error("tuning_get: unknown knob '%s'", name);
    return -2;
}

extern "C" int avs_persistent_cu_slots(void) { return avs_persistent_slots(); }
