// options.hip - the option surface of include/specmi.h: the table of every accepted name (default, stable / experimental), the
// gate on experimental names, and the set / get / info entry points.  Split out of api.hip in round 6.
#include <cmath>
#include <cstdarg>
#include <climits>
#include <cstdlib>
#include <cstring>

#include "handle.h"

using namespace specmi;

// ---- the option table: every name specmi_set_option_* accepts, its default and whether it is part of the STABLE surface
// (include/specmi.h).  Experimental names - tuning thresholds, debug pins, measured-slower opt-ins, the narrower-arithmetic
// secondary mode - are refused unless the process sets SPECMI_EXPERIMENTAL=1 or the handle's (stable) option "experimental" is 1;
// setting one to its default is always a no-op and allowed.  tests/test_abi.py checks the defaults here against the opt_i(...) call
// sites and against the header.  (0 = "by context" for persist_wgs / persist_fill_wgs.)
namespace {
struct OptSpec { const char* name; int def; bool stable; };
const OptSpec kOptions[] = {
    // stable: model shape (before commit)
    {"backbone", 50, true}, {"num_fc_layers", 1, true}, {"num_fc_channels", 1024, true}, {"use_cam", 0, true}, {"use_cam_feats", 0, true},
    {"img_res", 224, true}, {"hrnet_use_conv", 1, true}, {"estimate_var", 0, true}, {"uncertainty_activation", 0, true},
    // stable: execution (any time)
    {"plan", 0, true}, {"winograd", 1, true}, {"fuse_downsample", 1, true}, {"head_collapse", 1, true}, {"output_ld", 0, true},
    {"angle_ld", 0, true}, {"experimental", 0, true},
    // experimental: secondary arithmetic, debug pins, tuning thresholds, measured-slower or measured-neutral opt-ins
    {"conv_precision", 0, false}, {"conv_precision_3x3", 0, false}, {"force_conv_variant", 0, false}, {"force_wino_variant", 0, false},
    {"fc_splitk", 1, false}, {"fc_gemv", 1, false}, {"head_fuse", 3, false}, {"smpl_skin_split", -1, false},
    {"trunk_subbatch", 0, false}, {"trunk_subbatch_layers", 2, false},
    {"single_max_batch", 2, false}, {"latency_max_batch", 10, false}, {"latency_max_batch_single", 16, false},
    {"latency_target_wgs", 256, false}, {"latency_min_chunks", 4, false}, {"latency_wino_min_tiles", 128, false},
    {"latency_fill_wgs", 240, false}, {"latency_fill_wgs_large", 400, false}, {"latency_unit_model", 0, false},
    {"latency_unit_slots", 256, false}, {"latency_force_unit", 0, false},
    {"wsplit", 1, false}, {"wsplit_max_units", 1400, false}, {"wsplit_max_units_single", 500, false}, {"wsplit_slots", 256, false},
    {"conv2d_sk", 0, false}, {"conv2d_wsplit", 0, false},
    {"persist", 0, false}, {"persist_min_run", 2, false}, {"persist_max_run", 64, false}, {"persist_wgs", 0, false}, {"persist_fill_wgs", 0, false},
    {"persist_l2_prefetch", 0, false}, {"persist_spin_limit", 400000, false}, {"persist_allow_full", 0, false},
    {"tail_fuse", 0, false},
};
const OptSpec* find_option(const char* name) {
    for (const OptSpec& o : kOptions)
        if (std::strcmp(o.name, name) == 0) return &o;
    return nullptr;
}
bool experimental_allowed(specmi_handle* h) {
    const char* e = std::getenv("SPECMI_EXPERIMENTAL");
    if (e && e[0] && std::strcmp(e, "0") != 0) return true;
    auto it = h->opt_i.find("experimental");
    return it != h->opt_i.end() && it->second != 0;
}
}  // namespace

int opt_i(specmi_handle* h, const char* name, int dflt) {
    auto it = h->opt_i.find(name);
    return it == h->opt_i.end() ? dflt : it->second;
}
float opt_f(specmi_handle* h, const char* name, float dflt) {
    auto it = h->opt_f.find(name);
    return it == h->opt_f.end() ? dflt : it->second;
}

extern "C" {

int specmi_set_option_i32(specmi_handle* h, const char* name, int value) {
    if (!h || !name) return fail(h, SPECMI_ERR_ARG, "null argument");
    const OptSpec* o = find_option(name);
    if (!o) return fail(h, SPECMI_ERR_ARG, "unknown option '%s'", name);
    if (!o->stable && value != o->def && !experimental_allowed(h))
        return fail(h, SPECMI_ERR_STATE, "option '%s' is experimental (tuning / debug / measured-slower opt-in): set SPECMI_EXPERIMENTAL=1 "
                    "in the environment or option \"experimental\" = 1 on the handle first", name);
    h->opt_i[name] = value;
    return SPECMI_OK;
}

int specmi_set_option_f32(specmi_handle* h, const char* name, float value) {
    if (!h || !name) return fail(h, SPECMI_ERR_ARG, "null argument");
    if (std::strcmp(name, "focal_length") != 0) return fail(h, SPECMI_ERR_ARG, "unknown float option '%s'", name);
    h->opt_f[name] = value;
    return SPECMI_OK;
}

int specmi_get_option_i32(specmi_handle* h, const char* name, int* value) {
    if (!h || !name || !value) return fail(h, SPECMI_ERR_ARG, "null argument");
    const OptSpec* o = find_option(name);
    if (!o) return fail(h, SPECMI_ERR_ARG, "unknown option '%s'", name);
    auto it = h->opt_i.find(name);
    *value = it != h->opt_i.end() ? it->second : o->def;
    return SPECMI_OK;
}

int specmi_option_info(int index, const char** name, int* default_value, int* is_stable) {
    const int n = (int)(sizeof(kOptions) / sizeof(kOptions[0]));
    if (index < 0 || index >= n) return SPECMI_ERR_ARG;
    if (name) *name = kOptions[index].name;
    if (default_value) *default_value = kOptions[index].def;
    if (is_stable) *is_stable = kOptions[index].stable ? 1 : 0;
    return SPECMI_OK;
}

}  // extern "C"
