// config.h -- every switch of libomni_hip.so in ONE table (config.hip): name of the environment variable, default, valid range, class, one line
// of documentation.  Nothing else in the library reads the environment (omni_shard's OMNI_RCCL_LIB path is listed as the one string option).
//   * handles (omni_sp, omni_vlad, omni_index) take a snapshot with config_resolve() when they are CREATED and keep it: a value outside its range or
//     not a number fails the creation loudly instead of silently selecting something;
//   * launch-site hooks (traces, timing ablations, A/B forcing of a tile orientation, the index thresholds) are process-wide: config_process(), resolved
//     at FIRST USE -- the first launch or index search that looks at one of them -- and frozen from then on (loading the library freezes nothing: its
//     initializer reads OMNI_HW_QUEUES alone, config_option_now); omni_config_value reports the frozen value of such an option once it is frozen;
//   * include/omni_hip.h: omni_config_count / omni_config_describe / omni_config_value list the table and what a handle created now would see
//     (tests/test_config_cpu.py asserts the default variant set -- the production path -- from it).
#pragma once
#include "common.h"

namespace omni {

enum CfgId {
    // SuperPoint
    CFG_CONV_V1 = 0, CFG_CONV_RS, CFG_RS_TRN, CFG_DET16, CFG_SP_SPARSE_DESC, CFG_SP_SPARSE_DA, CFG_SP_FUSED_CAND, CFG_SP_SPLIT_DB, CFG_SP_MASK_SKIP, CFG_SP_MASK_SKIP_SPLIT, CFG_SPLIT_FUSE1A, CFG_SPLIT_WINO,
    CFG_SPLIT_TRN, CFG_CONV_XCD, CFG_SP_PROFILE_MASK, CFG_PP_U8, CFG_PP_TRACE, CFG_PP_DBG, CFG_RS_TRACE, CFG_SPLIT_TRACE, CFG_SPLIT_DBG, CFG_WINO_TRACE, CFG_ROCTX,
    // MobileNetVLAD
    CFG_VLAD_BIG, CFG_VLAD_STEM_FUSE, CFG_VLAD_UNFUSED, CFG_VLAD_MFMA, CFG_VLAD_SBLOCK, CFG_VLAD_MBLOCK_PX, CFG_VLAD_MFMA_PX, CFG_VLAD_FC_MFMA, CFG_VLAD_MBLOCK_CPW,
    CFG_VLAD_SB_LDSPAD, CFG_VLAD_SB_PERSIST, CFG_VLAD_SB_TRACE, CFG_VLAD_SB_DBG, CFG_VLAD_MASK_SKIP,
    // index
    CFG_SCAN_ROWS_MIN, CFG_MQ_ROT, CFG_MQ_MIN, CFG_INDEX_MIRROR, CFG_INDEX_MIRROR_MIN_ROWS, CFG_INDEX_CERT_FAIL,
    // host loop (libomni_host.so reads them through omni_config_value)
    CFG_GEOMETRY_THREADS, CFG_MESSAGE_THREADS, CFG_GEOMETRY_ASYNC, CFG_DETECTOR_ASYNC, CFG_PIPELINE_ONE_STREAM, CFG_PIPELINE_FIFO, CFG_PIPELINE_UNIT_PLAN,
    // runtime
    CFG_HW_QUEUES,
    // strings
    CFG_RCCL_LIB,
    CFG_COUNT
};
enum CfgClass {
    CFG_VARIANT = 0,    // another kernel / algorithm for the same results: A/B measurements and bit-identity tests; the default is the production path
    CFG_TUNING = 1,     // a threshold between two equivalent paths
    CFG_DEBUG = 2,      // traces and timing ablations (some give WRONG results: never in production)
    CFG_TEST = 3,       // fault injection for tests
    CFG_STRING = 4      // a path (no integer value)
};
struct CfgOption { const char* env; int def, lo, hi; int cls; const char* doc; };
extern const CfgOption kCfgOptions[CFG_COUNT];

struct Config {
    int v[CFG_COUNT];
    int operator[](CfgId i) const { return v[i]; }
};
// the table's defaults overridden by the environment AS IT IS NOW; OMNI_ERR_INVALID (+ omni_last_error) on a value that is not an integer of the option's range
int config_resolve(Config* out);
// resolved once, at first use (launch-site hooks); an invalid value there is reported on stderr and the default kept
const Config& config_process();
// ONE option parsed from the environment as it is now, nothing cached (the library's load-time initializer)
int config_option_now(CfgId i);

}  // namespace omni
