// odam_common.hip -- ABI-level helpers shared by all entry points: the error string and the switch table.
#include <atomic>
#include <cstdio>
#include <cstring>

#include "odam_config.h"
#include "odam_err.h"

extern "C" const char* odam_last_error(void) { return g_odam_err; }

namespace odam_cfg {
namespace {
struct Entry { const char* name; int def, lo, hi; };
const Entry TABLE[N_KEYS] = {
    {"cg.ring", 1, 0, 2},        {"cg.f32", 2, 0, 2},         {"cg.fuse", 2, 0, 2},       {"cg.fuse_bf16", 2, 0, 2},
    {"cg.s1", 1, 0, 1},          {"cg.ut", 1, 0, 1},          {"cg.tiles", 31, 0, 31},      {"cg.force", 0, 0, 3},
    {"cg.presplit", 1, 0, 1},    {"cg.mfma16", 3, 0, 3},      {"att.x3", 1, 0, 1},         {"att.bf16_mfma", 1, 0, 1}, {"stem.rows", 1, 0, 1},
    {"assoc.persist", 2, 0, 2},  {"sq.split", 32, 1, 32},     {"sq.split_wait", 3000, 0, 100000000}, {"sq.lanes", 0, 0, 64},
    {"stem.pool", 1, 0, 1},
    {"cg.pin", 0, 0, 1},
    {"cg.small_x3", 1, 0, 1},
    {"assoc.sk_fast", 1, 0, 3},
    {"assoc.hungarian", 0, 0, 1},
    {"assoc.merge", 1, 0, 1},
    {"sq.two_per_cu", 1, 0, 1},
};
std::atomic<int> g_val[N_KEYS];
std::atomic<bool> g_init{false};
void init() {
    if (g_init.load(std::memory_order_acquire)) return;
    for (int k = 0; k < N_KEYS; k++) g_val[k].store(TABLE[k].def, std::memory_order_relaxed);
    g_init.store(true, std::memory_order_release);
}
}  // namespace
int get(Key k) { init(); return g_val[k].load(std::memory_order_relaxed); }
void set(Key k, int v) { init(); g_val[k].store(v, std::memory_order_relaxed); }
}  // namespace odam_cfg

extern "C" int odam_config_set(const char* key, int value) {
    using namespace odam_cfg;
    if (!key) return odam_fail(1, "odam_config_set: null key");
    for (int k = 0; k < N_KEYS; k++)
        if (std::strcmp(key, TABLE[k].name) == 0) {
            if (value < TABLE[k].lo || value > TABLE[k].hi || (k == CG_F32 && value == 1)) {
                std::snprintf(g_odam_err, sizeof(g_odam_err), "odam_config_set: %s = %d outside %d..%d", key, value, TABLE[k].lo, TABLE[k].hi);
                return 1;
            }
            set((Key)k, value);
            return 0;
        }
    std::snprintf(g_odam_err, sizeof(g_odam_err), "odam_config_set: unknown key %s", key);
    return 1;
}

extern "C" int odam_config_get(const char* key, int* value) {
    using namespace odam_cfg;
    if (!key || !value) return odam_fail(1, "odam_config_get: null pointer");
    for (int k = 0; k < N_KEYS; k++)
        if (std::strcmp(key, TABLE[k].name) == 0) { *value = get((Key)k); return 0; }
    std::snprintf(g_odam_err, sizeof(g_odam_err), "odam_config_get: unknown key %s", key);
    return 1;
}

// "key=value key=value ..." of the whole table (values that differ from the default are followed by '*')
extern "C" int odam_config_describe(char* buf, int n) {
    using namespace odam_cfg;
    if (!buf || n <= 0) return odam_fail(1, "odam_config_describe: bad buffer");
    int pos = 0;
    buf[0] = 0;
    for (int k = 0; k < N_KEYS; k++) {
        const int v = get((Key)k);
        const int w = std::snprintf(buf + pos, (size_t)(n - pos), "%s%s=%d%s", k ? " " : "", TABLE[k].name, v, v != TABLE[k].def ? "*" : "");
        if (w < 0 || w >= n - pos) return odam_fail(1, "odam_config_describe: buffer too small");
        pos += w;
    }
    return 0;
}
