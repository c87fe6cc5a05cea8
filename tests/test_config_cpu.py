"""The library's switches are ONE table (csrc/config.h / config.hip; C entry points omni_config_*): this test pins the DEFAULT VARIANT SET -- the
production path -- and the table's behaviour: ranges enforced, nothing read from the environment outside it (a source scan), the reference variants
of the fp16 convolutions (OMNI_CONV_V1 = 1..3) absent from the shipped library's symbols and present in the test build."""
import glob
import os
import re
import subprocess

import pytest

import omni_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANT, TUNING, DEBUG, TEST, STRING = range(5)

# every variant switch and the value that ships: anything else is an A/B or test configuration
PRODUCTION = {
    "OMNI_CONV_V1": 0, "OMNI_CONV_RS": 2, "OMNI_RS_TRN": -1, "OMNI_DET16": 1, "OMNI_SP_SPARSE_DESC": 1, "OMNI_SP_SPARSE_DA": 1, "OMNI_SP_FUSED_CAND": 1, "OMNI_SP_SPLIT_DB": 1, "OMNI_SP_MASK_SKIP": 1,
    "OMNI_SP_MASK_SKIP_SPLIT": 1, "OMNI_SPLIT_FUSE1A": 1, "OMNI_SPLIT_WINO": 7, "OMNI_SPLIT_TRN": -1, "OMNI_CONV_XCD": 1, "OMNI_VLAD_STEM_FUSE": 1, "OMNI_VLAD_UNFUSED": 0, "OMNI_VLAD_MFMA": 1,
    "OMNI_VLAD_SBLOCK": 1, "OMNI_VLAD_FC_MFMA": 1, "OMNI_VLAD_SB_PERSIST": 1, "OMNI_VLAD_MASK_SKIP": 1, "OMNI_PP_U8": 1, "OMNI_MQ_ROT": 1, "OMNI_INDEX_MIRROR": 1, "OMNI_GEOMETRY_ASYNC": 1, "OMNI_DETECTOR_ASYNC": 1, "OMNI_PIPELINE_ONE_STREAM": 0, "OMNI_PIPELINE_FIFO": -1, "OMNI_PIPELINE_UNIT_PLAN": 1,
}


@pytest.fixture(scope="module")
def capi():
    return omni_loader.load().capi


def test_default_variant_set_is_the_production_path(capi):
    table = capi.config_table()
    names = [o["env"] for o in table]
    assert len(names) == len(set(names)) and all(n.startswith("OMNI_") for n in names)
    variants = {o["env"]: o["default"] for o in table if o["cls"] == VARIANT}
    assert variants == PRODUCTION
    for o in table:
        assert o["doc"], o["env"]
        if o["cls"] != STRING:
            assert o["lo"] <= o["default"] <= o["hi"], o
        if o["cls"] in (DEBUG, TEST):
            assert o["default"] == 0, o                         # traces, ablations and fault injection are off unless asked for
    tuning = {o["env"]: o["default"] for o in table if o["cls"] == TUNING}
    assert tuning["OMNI_INDEX_MIRROR_MIN_ROWS"] == 32768 and tuning["OMNI_MQ_MIN"] == 4 and tuning["OMNI_VLAD_MBLOCK_PX"] == 2048


def test_values_come_from_the_environment_and_ranges_are_enforced(capi, monkeypatch):
    for o in capi.config_table():
        if o["cls"] != STRING:
            monkeypatch.delenv(o["env"], raising=False)
    assert capi.config_value("OMNI_SP_SPARSE_DA") == 1
    monkeypatch.setenv("OMNI_SP_SPARSE_DA", "0")
    assert capi.config_value("OMNI_SP_SPARSE_DA") == 0
    for bad in ("2", "-1", "yes", "1x", "0.5"):
        monkeypatch.setenv("OMNI_SP_SPARSE_DA", bad)
        with pytest.raises(capi.OmniError, match="OMNI_SP_SPARSE_DA"):
            capi.config_value("OMNI_SP_SPARSE_DA")
        with pytest.raises(capi.OmniError, match="OMNI_SP_SPARSE_DA"):
            capi.config_value("OMNI_DET16")                     # ANY option asked for: a handle would refuse to be created
    monkeypatch.setenv("OMNI_SP_SPARSE_DA", "")
    assert capi.config_value("OMNI_SP_SPARSE_DA") == 1          # empty = unset
    monkeypatch.setenv("OMNI_SPLIT_TRN", "-1")
    assert capi.config_value("OMNI_SPLIT_TRN") == -1
    with pytest.raises(capi.OmniError, match="no option"):
        capi.config_value("OMNI_NO_SUCH_SWITCH")


def test_loading_the_library_freezes_no_process_wide_option(capi, monkeypatch):
    """the library's load-time initializer used to resolve the whole process-wide table (every launch-site hook and index threshold frozen at
    dlopen, before a caller could set anything).  It now reads OMNI_HW_QUEUES alone: a process-wide option set AFTER the library was loaded -- and before
    anything used one -- is what omni_config_value reports and what the first use will freeze (no GPU here: nothing has used one)."""
    capi.lib()
    for name, v in (("OMNI_CONV_XCD", "0"), ("OMNI_INDEX_MIRROR", "0"), ("OMNI_INDEX_MIRROR_MIN_ROWS", "12345")):
        monkeypatch.setenv(name, v)
        assert capi.config_value(name) == int(v)
        monkeypatch.delenv(name)
        assert capi.config_value(name) == {o["env"]: o["default"] for o in capi.config_table()}[name]


def test_nothing_reads_the_environment_outside_the_table(capi):
    names = {o["env"] for o in capi.config_table()}
    srcs = glob.glob(os.path.join(ROOT, "omni-swarm_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "omni-swarm_amd", "host", "*"))
    for f in srcs:
        text = open(f, errors="replace").read()
        for m in re.finditer(r'getenv\(\s*"?([A-Za-z0-9_.]*)"?', text):
            where = os.path.basename(f)
            assert where == "config.hip" or (where == "shard.hip" and m.group(1) == "OMNI_RCCL_LIB"), (where, m.group(0))
        for m in re.finditer(r'"(OMNI_[A-Z0-9_]+)"', text):
            if m.group(1) not in names:
                assert m.group(1) in ("OMNI_TEST_VARIANTS",) or not re.search(r"config_value|getenv", text[max(0, m.start() - 40):m.start()]), (os.path.basename(f), m.group(1))


def test_reference_conv_variants_are_only_in_the_test_build():
    nm = lambda lib: subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
    shipped = os.path.join(ROOT, "omni-swarm_amd", "lib", "libomni_hip.so")
    test = os.path.join(ROOT, "omni-swarm_amd", "lib_test", "libomni_hip.so")
    assert os.path.exists(test), "make -C omni-swarm_amd test-variants"
    v2 = "conv3x3_c64_f16_kernel"                               # the v2 persistent kernel; the generic fp16 3x3 instantiations go the same way
    generic16 = "conv_mfma_kernelIDF16_Li3E"
    assert v2 not in nm(shipped) and generic16 not in nm(shipped)
    assert v2 in nm(test) and generic16 in nm(test)


def test_process_wide_options_are_exactly_the_ones_read_through_config_process(capi):
    """config.hip keeps ONE list of the options frozen process-wide (what omni_config_value reports once frozen): it must be exactly the set the sources read
    through config_process()[...] -- an option read there but classed per-handle would report a value the kernels do not use."""
    used = set()
    for path in glob.glob(os.path.join(ROOT, "omni-swarm_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "omni-swarm_amd", "csrc", "*.h")):
        used |= set(re.findall(r"config_process\(\)\[(?:omni::)?(CFG_[A-Z0-9_]+)\]", open(path).read()))
    table = capi.config_table()
    ids = re.search(r"enum CfgId \{(.*?)CFG_COUNT", open(os.path.join(ROOT, "omni-swarm_amd", "csrc", "config.h")).read(), re.S).group(1)
    names = [n for n in re.findall(r"CFG_[A-Z0-9_]+", re.sub(r"//.*", "", ids))]
    assert len(names) == len(table)
    listed = {names[i] for i in range(len(names)) if capi.lib().omni_config_is_process_wide(i)}
    assert listed == used, (sorted(listed - used), sorted(used - listed))
