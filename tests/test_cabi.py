"""The C-ABI library loads and exports every symbol include/mdc.h declares (no GPU needed)."""
import ctypes
import os
import re

from conftest import ROOT
from modulationdetectioncnn_amd import _cabi


def _declared():
    text = open(os.path.join(ROOT, "include", "mdc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mdc_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared() == sorted(_cabi.EXPORTS)


def test_library_exports_every_declared_symbol():
    import modulationdetectioncnn_amd.build as b
    for variant in b.VARIANTS:          # the product library and the alternates test build: the same ABI
        lib = b.build(variant=variant)
        L = ctypes.CDLL(lib)
        for name in _declared():
            assert hasattr(L, name), (variant, name)
        assert L.mdc_abi_version() == _cabi.ABI_VERSION == 5


def test_dynamic_symbol_table_is_exactly_the_header():
    """Built with -fvisibility=hidden + a linker version script: `nm -D --defined-only` of either library lists the C
    entry points of include/mdc.h and NOTHING else -- no mangled mdc:: helpers, no libstdc++ template instantiations, no
    hipcc markers (35 C++ functions used to be exported next to the 25 C ones)."""
    import subprocess
    import modulationdetectioncnn_amd.build as b
    for variant in b.VARIANTS:
        out = subprocess.run(["nm", "-D", "--defined-only", b.build(variant=variant)], capture_output=True, text=True, check=True).stdout
        assert sorted(line.split()[-1] for line in out.splitlines() if line.strip()) == _declared(), variant


def test_trainer_entry_points_validate_their_arguments_without_gpu():
    L = _cabi.lib()
    L.mdc_last_error.restype = ctypes.c_char_p
    assert L.mdc_trainer_create(None, 0, None) == -22 and b"null" in L.mdc_last_error()
    assert L.mdc_train_batch(None, None, None, 1, None, 0, 1, 1, None) == -22 and b"null trainer" in L.mdc_last_error()
    assert L.mdc_trainer_evaluate(None, None, None, 1, None, 0, 1, None) == -22
    assert L.mdc_trainer_num_layers(None) == -22
    assert L.mdc_trainer_set_adam(None, 1e-3, 0.9, 0.999, 1e-7) == -22
    assert L.mdc_trainer_set_dropout(None, 0.5, 0) == -22
    assert L.mdc_trainer_read(None, 0, None, None, None, None, None, None) == -22
    L.mdc_trainer_destroy(None)


def test_product_library_has_one_kernel_per_role_and_never_reads_the_environment():
    """the measured-slower alternates (hipcc-scheduled bf16 conv, one-barrier dense1, the deployed nets'
    f32-MFMA dense layer) live in libmdc_alt.so only, and nothing in libmdc.so can call getenv -- the symbol is not even
    imported, so no entry point under mdc_forward* races a host thread's setenv."""
    import subprocess
    import modulationdetectioncnn_amd.build as b
    prod, alt = b.build(variant="product"), b.build(variant="alternates")
    undefined = subprocess.run(["nm", "-D", "--undefined-only", prod], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in undefined
    assert "getenv" in subprocess.run(["nm", "-D", "--undefined-only", alt], capture_output=True, text=True, check=True).stdout
    strings_prod = subprocess.run(["strings", prod], capture_output=True, text=True, check=True).stdout
    strings_alt = subprocess.run(["strings", alt], capture_output=True, text=True, check=True).stdout
    for env in ("MDC_DEP_PIVOT", "MDC_DEP_F32_MFMA", "MDC_CONV_SCHED", "MDC_DENSE1_PHASED", "MDC_D1_FUSED_HEAD"):
        assert env not in strings_prod, env
    alternates = ("deployed_f32m_kernel", "vt_conv_bf16_kernel", "vt_dense1_bf16_kernel")
    for k in alternates:
        assert not re.search(r"\d+%s[IE]" % k, strings_prod), k          # (mangled: <length><name>)
        assert re.search(r"\d+%s[IE]" % k, strings_alt), k
    for k in ("vt_conv_bf16_sched_kernel", "vt_dense1_bf16_phased_kernel", "deployed_fwd_kernel", "dense_chain_kernel"):
        assert re.search(r"\d+%s[IE]" % k, strings_prod), k


def test_binding_loads_and_reports_errors_without_gpu():
    L = _cabi.lib()
    L.mdc_last_error.restype = ctypes.c_char_p
    # null arguments are rejected before any device call
    assert L.mdc_create(None, 0, None) == -22
    assert b"null" in L.mdc_last_error()
    assert L.mdc_forward(None, None, 0, None, None, None, 0, None, 0, None) == -22
    assert L.mdc_workspace_bytes(None, 10) == 0


def test_new_entry_points_validate_their_arguments_without_gpu():
    """ABI v2 additions reject bad arguments before any device call (so this runs on a GPU-less host)."""
    L = _cabi.lib()
    L.mdc_last_error.restype = ctypes.c_char_p
    assert L.mdc_forward_iq_u8(None, None, 1, 128, 1.0, None, None, None, 0, None) == -22 and b"null model" in L.mdc_last_error()
    assert L.mdc_confusion_binned(None, None, None, -1, 3, 2, None, None, None) == -22
    assert L.mdc_confusion_binned(None, None, None, 5, 3, 2, None, None, None) == -22 and b"null labels" in L.mdc_last_error()
    one = (ctypes.c_int64 * 1)()
    assert L.mdc_confusion_binned(None, None, None, 0, 3, 0, ctypes.addressof(one), None, None) == -22 and b"bins" in L.mdc_last_error()
    assert L.mdc_confusion_binned(None, None, None, 0, 3, 4, ctypes.addressof(one), None, None) == 0       # n = 0: nothing to launch
    assert L.mdc_confusion(None, None, 0, 99, ctypes.addressof(one), None, None) == -22 and b"classes" in L.mdc_last_error()
    assert L.mdc_iq_u8_windows(None, 4, 0, 1.0, None, None) == -22 and b"hop" in L.mdc_last_error()
    assert L.mdc_iq_u8_windows(None, 4, 16, 1.0, None, None) == -22 and b"null buffer" in L.mdc_last_error()
    assert L.mdc_iq_u8_windows(None, 0, 16, 1.0, None, None) == 0
    buf = (ctypes.c_uint8 * 600)()                        # host memory is fine here: the alignment check comes before any launch
    assert L.mdc_iq_u8_windows(ctypes.addressof(buf) + 1, 2, 16, 1.0, ctypes.addressof(buf) + 8, None) == -22 and b"2-byte" in L.mdc_last_error()
    topo = _cabi.MdcTopology(1, 3, 0, 3, (ctypes.c_int32 * 4)(2, 0, 0, 0))       # an option bit the ABI does not define
    h = ctypes.c_void_p()
    assert L.mdc_create(ctypes.byref(topo), 0, ctypes.byref(h)) == -22 and b"option" in L.mdc_last_error()
    topo = _cabi.MdcTopology(1, 3, 0, 3, (ctypes.c_int32 * 4)(0, 0, 1, 0))
    assert L.mdc_create(ctypes.byref(topo), 0, ctypes.byref(h)) == -22 and b"reserved" in L.mdc_last_error()
    assert L.mdc_iq_u8_to_frames(None, -1, 1.0, None, None) == -22
    assert L.mdc_forward_q612(None, None, 0, 1, None, None, None) == -22
    assert L.mdc_set_fp8_input_absmax(None, 1.0) == -22
    assert L.mdc_profile_read(None, 0, None, None) == -22 and L.mdc_profile_reset(None) == -22 and L.mdc_set_profiling(None, 1) == -22
    assert L.mdc_predict_host(None, None, 4, None, None, 0) == -22 and b"null model" in L.mdc_last_error()
    assert L.mdc_predict_host_iq_u8(None, None, 4, 128, 1.0, None, None, 0) == -22 and b"null model" in L.mdc_last_error()
    L.mdc_destroy(None)                                                                                   # a no-op, not a crash


def test_header_is_plain_c99(tmp_path):
    """include/mdc.h compiles as C99 with -Wall -Werror -pedantic and nothing but the standard headers."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "mdc.h"\nint main(void) { return MDC_ABI_VERSION == 5 && MDC_HOP_FRAME == 128 && MDC_OPT_ALL == 1 && MDC_OPT_FP8_BF16_FEATURES == 1 ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(tmp_path / "hdr")], check=True)
    assert subprocess.run([str(tmp_path / "hdr")]).returncode == 0


def test_build_cache_is_keyed_on_flags(tmp_path, monkeypatch):
    """a build with other flags (a -DMDC_ABLATIONS probe build) must not leave objects for a plain build."""
    import hashlib
    import modulationdetectioncnn_amd.build as b
    key = hashlib.sha256(" ".join(b.CXXFLAGS).encode()).hexdigest()
    b.build()
    assert open(os.path.join(b.OBJ, "flags.stamp")).read().strip() == key == b.flag_key()
    assert b.flag_key(["-DMDC_ABLATIONS"]) != key
    # the alternates build keeps its own objects, stamp and library: neither build can pick up the other's
    obj_alt, lib_alt, flags_alt = b.VARIANTS["alternates"]
    b.build(variant="alternates")
    assert obj_alt != b.OBJ and lib_alt != b.LIB and open(os.path.join(obj_alt, "flags.stamp")).read().strip() == b.flag_key(flags_alt) != key


def test_build_drops_objects_without_a_source():
    """csrc/build/vtcnn2_fp8_pc.o (a removed kernel file's object) kept travelling to the GPU box."""
    import modulationdetectioncnn_amd.build as b
    stale = os.path.join(b.OBJ, "removed_kernel_file.o")
    open(stale, "wb").close()
    b.build()
    assert not os.path.exists(stale)
    assert sorted(f[:-2] for f in os.listdir(b.OBJ) if f.endswith(".o")) == sorted(os.path.basename(s)[:-4] for s in b.sources())


def test_no_oracle_import_in_product():
    pkg = os.path.join(ROOT, "modulationdetectioncnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "oracle_np" not in src, f
