"""The asm-sequenced conv kernel relies on an instruction ORDER for hazards the compiler does not model.  One of
them depends on register allocation (store operands reused as the destination of a following VGPR-writing MFMA,
tools/lint_async_hazards.py): check the generated ISA on every build.  CPU only (hipcc cross-compiles)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("inst", ["ILi0ELb0ELb0", "ILi0ELb1ELb0", "ILi0ELb0ELb1", "ILi0ELb1ELb1"])      # <ABL 0, U8, RANGE>
def test_sched_kernel_has_no_store_vs_mfma_hazard(inst):
    """every shipped instantiation: f32 frames / raw bytes, batch form / position-range form (small batches)"""
    spec = importlib.util.spec_from_file_location("lint_async_hazards", os.path.join(ROOT, "tools", "lint_async_hazards.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    isa = lint.kernel_isa(os.path.join(ROOT, "modulationdetectioncnn_amd", "csrc", "vtcnn2_bf16_sched.hip"), "vt_conv_bf16_sched_kernel" + inst)
    assert sum(1 for x in isa if x.startswith("v_mfma")) > 1000          # the kernel was found and is unrolled
    assert sum(1 for x in isa if x.startswith("global_store")) >= 30          # (one dwordx2 per step, one dword per odd step, the tails)
    assert lint.lint(isa) == []


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("inst", ["ILb0ELb0ELb1", "ILb1ELb0ELb1", "ILb0ELb1ELb1", "ILb1ELb1ELb1",      # <U8, RANGE, F8OUT = E4M3 features>
                                  "ILb0ELb0ELb0", "ILb1ELb0ELb0", "ILb0ELb1ELb0", "ILb1ELb1ELb0"])     # ... bf16 features (MDC_OPT_FP8_BF16_FEATURES)
def test_fp8_kernel_has_no_store_vs_mfma_hazard(inst):
    spec = importlib.util.spec_from_file_location("lint_async_hazards", os.path.join(ROOT, "tools", "lint_async_hazards.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    isa = lint.kernel_isa(os.path.join(ROOT, "modulationdetectioncnn_amd", "csrc", "vtcnn2_fp8_conv.hip"), "vt_conv_fp8_kernel" + inst)
    assert sum(1 for x in isa if x.startswith("v_mfma_scale")) > 300
    assert lint.lint(isa) == []


def test_lint_flags_the_pattern():
    spec = importlib.util.spec_from_file_location("lint_async_hazards", os.path.join(ROOT, "tools", "lint_async_hazards.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    bad = ["global_store_dwordx2 v[136:137], v[134:135], off", "v_mfma_f32_16x16x16_bf16 v[134:137], v[164:165], v[176:177], 0"]
    assert len(lint.lint(bad)) == 1
    ok = ["global_store_dwordx2 v[136:137], v[134:135], off", "v_mfma_f32_16x16x32_bf16 a[0:3], v[164:167], v[176:179], a[0:3]"]
    assert lint.lint(ok) == []
    mov = ["v_accvgpr_mov_b32 a200, a160", "v_mfma_f32_16x16x32_bf16 a[200:203], a[80:83], v[180:183], a[200:203]"]
    assert len(lint.lint(mov)) == 1
    padded = ["v_accvgpr_mov_b32 a200, a160", "s_nop 1", "v_mfma_f32_16x16x32_bf16 a[200:203], a[80:83], v[180:183], a[200:203]"]
    assert lint.lint(padded) == []
    short = ["v_accvgpr_mov_b32 a200, a160", "s_nop 0", "v_mfma_f32_16x16x32_bf16 a[200:203], a[80:83], v[180:183], a[200:203]"]
    assert len(lint.lint(short)) == 1
    waited = ["ds_write_b128 v10, v[20:23]", "s_waitcnt lgkmcnt(0)", "v_mfma_f32_16x16x16_bf16 v[20:23], v[1:2], v[3:4], 0"]
    assert lint.lint(waited) == []


def _tool(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("inst", ["ILi0ELb1ELb1", "ILi0ELb0ELb1"])      # <ABL 0, HEAD, F8 = true>
def test_dense1_fp8_mode_reads_its_fragments_with_plain_ds_read_b64(inst):
    """hipcc merged half of the E4M3 fragment reads into ds_read2st64_b64, whose 4 x 16-lane / 32-bank
    service the source swizzle was not derived for (LDS bank conflicts 0.29 of the LDS-active cycles).  They are asm
    ds_read_b64 now; no two-address 64-bit read may come back."""
    lint = _tool("lint_async_hazards")
    isa = lint.kernel_isa(os.path.join(ROOT, "modulationdetectioncnn_amd", "csrc", "vtcnn2_bf16_dense1.hip"), "vt_dense1_bf16_phased_kernel" + inst)
    assert sum(1 for x in isa if x.startswith("v_mfma_f32_16x16x32")) >= 64
    assert not [x for x in isa if x.startswith(("ds_read2_b64", "ds_read2st64_b64"))]
    assert sum(1 for x in isa if x.startswith("ds_read_b64")) == 32       # prologue + phase 2 + phase 3 (two code paths): 8 each
    assert lint.lint(isa) == []


def test_no_product_kernel_uses_scratch():
    """Every kernel of the BUILT libmdc.so: .private_segment_fixed_size == 0 and no VGPR spill (round 4 shipped
    deployed_q612_kernel<10> with 22 VGPRs in scratch).  Read from the library's own code objects, not from a recompile."""
    import modulationdetectioncnn_amd.build as b
    meta = _tool("kernel_meta").kernel_metadata(b.build(variant="product"))
    assert len(meta) >= 60
    assert any("deployed_q612_kernel" in k for k in meta) and any("train_deployed_kernel" in k for k in meta)
    bad = {k: r for k, r in meta.items() if r["scratch"] != 0 or r["vgpr_spill"] != 0}
    assert not bad, bad


def _serialised_lds_mfma(isa):
    """Count `ds_read*; s_waitcnt lgkmcnt(0); v_mfma*` triples: an MFMA stalled on an LDS read issued just before it."""
    return sum(1 for a, b, c in zip(isa, isa[1:], isa[2:]) if a.startswith("ds_read") and b.startswith("s_waitcnt") and "lgkmcnt(0)" in b
               and c.startswith("v_mfma"))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
@pytest.mark.parametrize("src,kernel,chain_mfmas", [("dense_chain.hip", "dense_chain_kernelILi2ELi3ELb0E", 128), ("dense_chain.hip", "dense_chain_kernelILi1ELi1ELb0E", 64),
                                                    ("train.hip", "train_cnnpy_kernelILb1ELb0E", 256)])
def test_layer_1_mfma_chains_do_not_wait_out_their_lds_reads(src, kernel, chain_mfmas):
    """Round 5: hipcc read each pair of layer-1 A operands into the SAME two registers right after the MFMAs that consumed the last
    pair, then waited `lgkmcnt(0)` in front of the next group -- 31 exposed LDS round trips per 16-row tile in cnn.py's net.  The
    loops request the next pair before the current MFMAs issue (pinned with sched_barrier): in the ISA the long chains must wait
    with a COUNTED lgkmcnt, and a read directly in front of `lgkmcnt(0)` + MFMA may only remain in the short tail layers."""
    lint = _tool("lint_async_hazards")
    isa = lint.kernel_isa(os.path.join(ROOT, "modulationdetectioncnn_amd", "csrc", src), kernel)
    assert sum(1 for x in isa if x.startswith("v_mfma_f32_16x16x4")) >= chain_mfmas
    counted = sum(1 for a, b in zip(isa, isa[1:]) if a.startswith("s_waitcnt") and "lgkmcnt(1)" in a and b.startswith("v_mfma"))
    assert counted >= chain_mfmas // 8, counted                     # the pipelined groups wait for the OLDER read only
    assert _serialised_lds_mfma(isa) <= 10, _serialised_lds_mfma(isa)      # (before: 31 in layer 1 alone)
