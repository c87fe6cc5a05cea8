// config.hip -- the table behind config.h and its C entry points (include/omni_hip.h: omni_config_*)
#include "config.h"

#include <atomic>
#include <cerrno>
#include <climits>
#include <cstdlib>

namespace omni {

const CfgOption kCfgOptions[CFG_COUNT] = {
    // ---- SuperPoint ---------------------------------------------------------------------------------------------------------------------------------
    {"OMNI_CONV_V1", 0, 0, 3, CFG_VARIANT, "fp16 3x3 layers: 0 = production (conv1a fused into the ping-pong conv1b, register-stationary cin=128); 1 = generic kernel, 2 = persistent LDS-DMA kernel, "
                                           "3 = ping-pong without the conv1a fusion -- 1-3 exist only in the test build of the library (lib_test/)"},
    {"OMNI_CONV_RS", 2, 0, 2, CFG_VARIANT, "cin=128 fp16 layers: 2 = the register-stationary kernel with its epilogue and DMA inside the MFMA stream (3 / 4-row tiles), 1 = the 6-row-tile "
                                          "register-stationary kernel of rounds 1-5, 0 = generic kernel"},
    {"OMNI_RS_TRN", -1, -1, 1, CFG_VARIANT, "register-stationary kernel: tile orientation, -1 = the one with fewer tiles, 0 = plain, 1 = transposed"},
    {"OMNI_DET16", 1, 0, 1, CFG_VARIANT, "detector head of the fp16 and OMNI_PREC_SPLIT paths on v_mfma_f32_32x32x16_f16 with split (hi, lo) operands (0: the exact-f32 MFMA kernel)"},
    {"OMNI_SP_SPARSE_DESC", 1, 0, 1, CFG_VARIANT, "convDb + descriptor norm only at the cells around the key points (0: dense descriptor map)"},
    {"OMNI_SP_SPARSE_DA", 1, 0, 1, CFG_VARIANT, "convDa only at those cells too (fp16 and OMNI_PREC_SPLIT; 0: dense convDa)"},
    {"OMNI_SP_FUSED_CAND", 1, 0, 1, CFG_VARIANT, "getKeyPoints' threshold inside the detector head's epilogue + window masks from the candidate list (0: sp_cand_kernel re-reads the heat map)"},
    {"OMNI_SP_SPLIT_DB", 1, 0, 1, CFG_VARIANT, "OMNI_PREC_SPLIT: convDb + descriptor norm at the key points' cells with split (hi, lo) operands on the fp16 matrix cores (0: exact-f32 MFMA convolution)"},
    {"OMNI_SP_MASK_SKIP", 1, 0, 1, CFG_VARIANT, "fp16: the tiles inside the constant region of the fisheye mask are left out of the tile walk (0: every tile)"},
    {"OMNI_SP_MASK_SKIP_SPLIT", 1, 0, 1, CFG_VARIANT, "the same for OMNI_PREC_SPLIT"},
    {"OMNI_SPLIT_FUSE1A", 1, 0, 1, CFG_VARIANT, "OMNI_PREC_SPLIT: conv1a built inside the conv1b kernel from the u8 image (0: separate exact-f32 conv1a pass)"},
    {"OMNI_SPLIT_WINO", 7, 0, 15, CFG_VARIANT, "OMNI_PREC_SPLIT: the cin = 64 layers as Winograd F(2x2,3x3) kernels with split operands (conv_wino.hip), bit 0 = conv1b (with the conv1a "
                                                "fusion), 1 = conv2a, 2 = conv2b, 3 = conv3a (off by default: same time as the direct kernel, profiles/r06g); 0: the direct kernels of conv_split.hip"},
    {"OMNI_SPLIT_TRN", -1, -1, 1, CFG_VARIANT, "OMNI_PREC_SPLIT cin=128 kernel: tile orientation, as OMNI_RS_TRN"},
    {"OMNI_CONV_XCD", 1, 0, 1, CFG_VARIANT, "persistent convolution kernels derive (cout group, tile walk) from an XCD-aware block id: the cout groups of a pixel tile and its neighbours "
                                           "share one L2 (0: plain blockIdx; process-wide, same results either way)"},
    {"OMNI_SP_PROFILE_MASK", 0, 0, 1, CFG_TUNING, "omni_sp_profile times the stages with the fisheye mask on (what the key-frame pipeline runs)"},
    {"OMNI_PP_U8", 1, 0, 1, CFG_VARIANT, "fp16 conv1a inside conv1b's kernel: matrix-core operands straight from the image bytes (0: through the 256-entry u8 -> (hi, lo) table in LDS)"},
    {"OMNI_PP_TRACE", 0, 0, 1, CFG_DEBUG, "s_memtime trace of the ping-pong conv kernel on stderr"},
    {"OMNI_PP_DBG", 0, 0, 255, CFG_DEBUG, "ping-pong conv kernel timing ablations (WRONG results)"},
    {"OMNI_RS_TRACE", 0, 0, 1, CFG_DEBUG, "s_memtime trace of the register-stationary kernel"},
    {"OMNI_SPLIT_TRACE", 0, 0, 1, CFG_DEBUG, "s_memtime trace of the split-precision kernel"},
    {"OMNI_SPLIT_DBG", 0, 0, 3, CFG_DEBUG, "split-precision kernel timing ablations: 1 = no stores, 2 = every DMA reads tile 0 (WRONG results)"},
    {"OMNI_WINO_TRACE", 0, 0, 1, CFG_DEBUG, "s_memtime trace of the Winograd split kernel"},
    {"OMNI_ROCTX", 0, 0, 1, CFG_DEBUG, "roctx ranges around upload / SuperPoint / MobileNetVLAD / post-processing / BF match / index add + search / exchange and the host loop's "
                                      "stages (rocprofv3 --marker-trace; the library is resolved by dlopen)"},
    // ---- MobileNetVLAD ------------------------------------------------------------------------------------------------------------------------------
    {"OMNI_VLAD_BIG", 0, 0, 128, CFG_TUNING, "64 / 128: the 64- / 128-pixel tiles of the unfused block kernel (measured slower at 600x480)"},
    {"OMNI_VLAD_STEM_FUSE", 1, 0, 1, CFG_VARIANT, "stem + block 0 in one kernel (0: two kernels)"},
    {"OMNI_VLAD_UNFUSED", 0, 0, 1, CFG_VARIANT, "1: every block as expand / depthwise / project launches"},
    {"OMNI_VLAD_MFMA", 1, 0, 1, CFG_VARIANT, "late blocks' 1x1 convolutions on the matrix cores"},
    {"OMNI_VLAD_SBLOCK", 1, 0, 1, CFG_VARIANT, "fused split-fp16 block kernel (vlad_s.hip)"},
    {"OMNI_VLAD_MBLOCK_PX", 2048, 0, 1 << 24, CFG_TUNING, "fused matrix-core block kernel for blocks of at most this many input pixels per image (0 disables)"},
    {"OMNI_VLAD_MFMA_PX", 0, 0, 1 << 24, CFG_TUNING, "> 0: matrix-core 1x1 convolutions for blocks of at most this many input pixels per image (default 2048)"},
    {"OMNI_VLAD_FC_MFMA", 1, 0, 1, CFG_VARIANT, "the final FC on the matrix cores (0: VALU kernel)"},
    {"OMNI_VLAD_MBLOCK_CPW", 0, 0, 64, CFG_TUNING, "hidden-layer split of the matrix-core block kernel: chunks per workgroup (0 = no split; measured: does not pay)"},
    {"OMNI_VLAD_SB_LDSPAD", 0, 0, 160 * 1024, CFG_DEBUG, "extra LDS bytes per workgroup of the split block kernel (occupancy A/B)"},
    {"OMNI_VLAD_SB_PERSIST", 1, 0, 16, CFG_VARIANT, "split block kernel: 0 = one tile per workgroup, N >= 1 = N x (CUs x resident workgroups) persistent workgroups"},
    {"OMNI_VLAD_SB_TRACE", 0, 0, 1, CFG_DEBUG, "s_memtime trace of the split block kernel"},
    {"OMNI_VLAD_SB_DBG", 0, 0, 255, CFG_DEBUG, "split block kernel timing ablations (WRONG results)"},
    {"OMNI_VLAD_MASK_SKIP", 1, 0, 1, CFG_VARIANT, "fisheye-masked passes leave the tiles of the first blocks that lie in the constant region of the mask out of the tile walk "
                                                  "(bit-identical; 0: the dense pass)"},
    // ---- index -----------------------------------------------------------------------------------------------------------------------------------------
    {"OMNI_SCAN_ROWS_MIN", 4, 1, 1 << 20, CFG_TUNING, "fp32 scan: from this many queries on, the rows-stationary kernel"},
    {"OMNI_MQ_ROT", 1, 0, 1, CFG_VARIANT, "matrix-core multi-query scan: rotated query fragments"},
    {"OMNI_MQ_MIN", 4, 0, 1 << 20, CFG_TUNING, "queries from which an fp16 shard is searched on the matrix cores (1 = always, 0 = never)"},
    {"OMNI_INDEX_MIRROR", 1, 0, 1, CFG_VARIANT, "fp32 index: fp16 mirror + certificate for batched searches (0: exact scans only)"},
    {"OMNI_INDEX_MIRROR_MIN_ROWS", 32768, 0, INT_MAX, CFG_TUNING, "the mirror is used from this many rows on"},
    {"OMNI_INDEX_CERT_FAIL", 0, 0, 1, CFG_TEST, "1: every certificate fails (exercises the exact fallback)"},
    // ---- host loop -------------------------------------------------------------------------------------------------------------------------------------
    {"OMNI_GEOMETRY_THREADS", -1, -1, 1024, CFG_TUNING, "threads of the geometric-verification pool (-1: min(16, cores / 2); 0: inline)"},
    {"OMNI_MESSAGE_THREADS", 3, 0, 64, CFG_TUNING, "helper threads that fill a unit's key-frame messages (key points, descriptors, lifted points: ~75 KB per image, first-touch page "
                                                  "faults included) next to the calling thread while the unit's detector step runs on the GPU (0: inline)"},
    {"OMNI_GEOMETRY_ASYNC", 1, 0, 1, CFG_VARIANT, "a micro-batch's geometry tasks run while the next unit is waited for (0: drained at once)"},
    {"OMNI_DETECTOR_ASYNC", 1, 0, 1, CFG_VARIANT, "a micro-batch's detector step (appends, searches) is enqueued and collected one unit later (0: the host waits for it on the spot)"},
    {"OMNI_PIPELINE_ONE_STREAM", 0, 0, 1, CFG_VARIANT, "a unit's MobileNetVLAD launches behind its SuperPoint launches on one stream (0: next to them on a second stream)"},
    {"OMNI_PIPELINE_FIFO", -1, -1, 2, CFG_VARIANT, "units in flight run oldest first: a unit's SuperPoint stream (1) / both its streams (2) start behind the convolution stack of the unit "
                                                   "enqueued before it; 0: the units' kernels take turns; -1: by measurement -- 1 for the fp32-class precisions, whose time is all "
                                                   "CU-filling convolutions (+2-4 %), and for an fp16 run() of no more units than lanes (all in flight at once: +13 % at 3 units), 0 for fp16 otherwise, whose small-grid "
                                                   "tails the next units' kernels fill (-5 % when chained)"},
    {"OMNI_PIPELINE_UNIT_PLAN", 1, 0, 2, CFG_VARIANT, "KeyframePipeline::run on host blocks, a run that is not a whole number of micro-batches: 1 = units of equal size (20 key frames = "
                                                       "7 + 7 + 6), 2 = the same behind half a unit, 0 = the blocks' own cut (8 + 8 + 4)"},
    // ---- runtime ---------------------------------------------------------------------------------------------------------------------------------------
    {"OMNI_HW_QUEUES", 8, 0, 64, CFG_TUNING, "hardware queues asked of the HIP runtime when the library is loaded (GPU_MAX_HW_QUEUES, unless already set): the pipeline's five "
                                            "streams must not share one; 0 = the runtime's default of 4"},
    // ---- strings ---------------------------------------------------------------------------------------------------------------------------------------
    {"OMNI_RCCL_LIB", 0, 0, 0, CFG_STRING, "path of the RCCL library omni_shard dlopens (default: librccl.so next to torch, then the loader's search path)"},
};

static int parse_one(const CfgOption& o, int* out, bool* bad) {
    *out = o.def; *bad = false;
    if (o.cls == CFG_STRING) return 0;
    const char* e = getenv(o.env);
    if (!e || !e[0]) return 0;
    errno = 0;
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (errno || end == e || *end != '\0' || v < o.lo || v > o.hi) { *bad = true; return 0; }
    *out = (int)v;
    return 1;
}

int config_resolve(Config* out) {
    for (int i = 0; i < CFG_COUNT; ++i) {
        bool bad;
        parse_one(kCfgOptions[i], &out->v[i], &bad);
        OMNI_REQUIRE(!bad, OMNI_ERR_INVALID, "%s=%s: expected an integer in [%d, %d] (%s)", kCfgOptions[i].env, getenv(kCfgOptions[i].env), kCfgOptions[i].lo, kCfgOptions[i].hi,
                     kCfgOptions[i].doc);
    }
    return OMNI_OK;
}

// the options read through config_process(): frozen for the process when the first of them is looked at (launch-site hooks, index thresholds).  ONE list, used by
// omni_config_value (what it reports once frozen) and exported (omni_config_is_process_wide): tests/test_config_cpu.py scans the sources for
// config_process()[...] call sites and asserts that they are exactly this list, so it cannot drift from them unnoticed.
static bool is_process_wide(int i) {
    switch (i) {
        case CFG_CONV_RS: case CFG_CONV_XCD: case CFG_INDEX_CERT_FAIL: case CFG_INDEX_MIRROR: case CFG_INDEX_MIRROR_MIN_ROWS: case CFG_MQ_ROT:
        case CFG_PP_DBG: case CFG_PP_TRACE: case CFG_RS_TRACE: case CFG_RS_TRN: case CFG_SCAN_ROWS_MIN: case CFG_SPLIT_DBG: case CFG_SPLIT_TRACE: case CFG_SPLIT_TRN: case CFG_WINO_TRACE: case CFG_ROCTX:
        case CFG_VLAD_BIG: case CFG_VLAD_SB_DBG: case CFG_VLAD_SB_LDSPAD: case CFG_VLAD_SB_TRACE:
            return true;
        default:
            return false;
    }
}
static std::atomic<bool> g_process_resolved{false};

int config_option_now(CfgId i) {
    int v; bool bad;
    parse_one(kCfgOptions[i], &v, &bad);
    return v;                                   // (an invalid value: the default)
}

const Config& config_process() {
    static const Config c = [] {
        g_process_resolved.store(true);
        Config q;
        for (int i = 0; i < CFG_COUNT; ++i) {
            bool bad;
            parse_one(kCfgOptions[i], &q.v[i], &bad);
            if (bad) fprintf(stderr, "libomni_hip: %s=%s ignored: expected an integer in [%d, %d]\n", kCfgOptions[i].env, getenv(kCfgOptions[i].env), kCfgOptions[i].lo, kCfgOptions[i].hi);
        }
        return q;
    }();
    return c;
}

}  // namespace omni

extern "C" {

int omni_config_count(void) { return omni::CFG_COUNT; }

int omni_config_describe(int i, const char** env, int* def, int* lo, int* hi, int* cls, const char** doc) {
    OMNI_REQUIRE(i >= 0 && i < omni::CFG_COUNT, OMNI_ERR_INVALID, "omni_config_describe: option %d of %d", i, (int)omni::CFG_COUNT);
    const omni::CfgOption& o = omni::kCfgOptions[i];
    if (env) *env = o.env;
    if (def) *def = o.def;
    if (lo) *lo = o.lo;
    if (hi) *hi = o.hi;
    if (cls) *cls = o.cls;
    if (doc) *doc = o.doc;
    return OMNI_OK;
}

int omni_config_is_process_wide(int i) { return (i >= 0 && i < omni::CFG_COUNT && omni::is_process_wide(i)) ? 1 : 0; }

int omni_config_value(const char* env, int* value) {
    OMNI_REQUIRE(env && value, OMNI_ERR_INVALID, "null argument");
    omni::Config c;
    const int rc = omni::config_resolve(&c);
    if (rc) return rc;
    for (int i = 0; i < omni::CFG_COUNT; ++i)
        if (!strcmp(omni::kCfgOptions[i].env, env)) {
            // a process-wide option that is already frozen: what the kernels really use, whatever the environment says by now
            *value = (omni::is_process_wide(i) && omni::g_process_resolved.load()) ? omni::config_process()[(omni::CfgId)i] : c.v[i];
            return OMNI_OK;
        }
    omni::set_error("omni_config_value: no option named %s", env);
    return OMNI_ERR_INVALID;
}

}  // extern "C"
