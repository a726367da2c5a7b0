"""Guard rail for the exactness claim of DESIGN.md §3: the reference sums `(a-b)^2` / `a*b` with
SEPARATELY rounded multiplies and adds (Rust never contracts: rs/utils/src/distance/l2.rs:32-89, dot_product.rs:38-71,
rs/quantization/src/pq/mod.rs:202-278), so the kernels that carry the reference's association must not contain a fused
multiply-add.  `-ffp-contract=off` and the pragma are the compile-time guards; this test disassembles the gfx950 code
objects inside the built libmuopdb_hip.so (llvm-objdump, no GPU needed) and asserts it.  Kernels that are NOT on the
exact path (the MFMA filter and its norm / prep kernels, the k-means update, column means) are allowed to fuse.

Two fused forms are legitimate inside the exact kernels and are recognised by shape, not whitelisted by count:
  * the Newton residuals of the correctly rounded `sqrtf` (`v_fma_f32 r, -s, s, x` right after v_sqrt / v_rsq: a NEGATED
    operand — an accumulate step `acc += d * d` never has one);
  * the u64 <-> f32 conversions of integer division (`v_fmamk_f32 .., 0x4f800000 / 0xcf800000` = +-2^32, or a literal 0
    multiplier, next to v_rcp)."""
import os
import re
import shutil
import subprocess

import pytest

from muopdb_amd import lib as L

LLVM = "/opt/rocm/lib/llvm/bin"
FUSED = re.compile(r"\b(v_fma_f|v_fmac_f|v_mad_f|v_mac_f|v_pk_fma|v_pk_mad|v_fmaak|v_fmamk|v_madak|v_madmk|v_dot\d|v_mfma)")
# kernels whose results carry the reference's lane association (every distance that is RETURNED or RANKED exactly)
EXACT = re.compile(r"(flat_scan_kernel|flat_small_scan_kernel|flat_small_block_kernel|flat_refine_kernel|ivf_scan_f32_kernel|ivf_scan_pq2?_kernel|ivf_pq3_refine_kernel|ivf_pq_fused_kernel|ivf_prep_kernel|ivf_coarse_rank_kernel|merge_rows_remap_kernel|"
                   r"hnsw_(beam|search|closure|pipe|select)_kernel|hnsw_upper_(top|top_rank|table\w*)_kernel|"
                   r"pair_distance_kernel|lane_conforming_kernel|pq_quantize_kernel|pq_distance_kernel|pq_rows_kernel|spann_filter_kernel|"
                   r"kmeans_assign_kernel)")
# kernels whose hot loops must address LDS as LDS: a `volatile` access through a generic pointer is never rewritten to the LDS
# address space and compiles to flat_load / flat_store + s_waitcnt vmcnt(0) (round 2: the pipelined HNSW kernel's mailbox polls;
# a non-inlined lambda did the same to the PQ table reads).  The HNSW kernels keep the flat accesses of their shared fallback
# (hnsw_general_traverse: the visited set may live in HBM there), so for those the bound is the beam kernel's own count.
NO_FLAT = re.compile(r"(ivf_scan_pq3_kernel|ivf_pq3_refine_kernel|ivf_scan_pq2_kernel|ivf_scan_f32_kernel|flat_scan_kernel|flat_bf16_filter_kernel|flat_bf16x1_block_kernel|"
                     r"flat_refine_kernel|sample_bound_kernel)")


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="llvm-objdump not installed")
def test_exact_kernels_contain_no_fused_multiply_add(tmp_path):
    so = tmp_path / "libmuopdb_hip.so"
    shutil.copy(L.LIB_PATH, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so.name], cwd=tmp_path, check=True, capture_output=True)
    objs = [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]
    assert objs, "no gfx950 code object inside libmuopdb_hip.so"
    kernels, offenders, explained = 0, [], 0
    for f in objs:
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        cur = None
        seen = set()
        recent = []  # mnemonics of the last instructions of the current function
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1) if EXACT.search(m.group(1)) else None
                recent = []
                if cur and cur not in seen:
                    seen.add(cur)
                    kernels += 1
                continue
            ins = line.split("//")[0].strip()
            if not cur or not ins:
                continue
            if FUSED.search(ins):
                # (a table kernel finishes the square roots of several queries' distances together: their refinement steps interleave, so a
                # residual sits further behind its v_sqrt; the NEGATED operand is what tells a residual from a contracted accumulate)
                near_sqrt = any(r.startswith(("v_sqrt_f32", "v_rsq_f32")) for r in recent[-160:])
                near_rcp = any(r.startswith("v_rcp_") for r in recent[-32:])
                sqrt_residual = ins.startswith("v_fma_f32") and re.search(r", -v\d+", ins) and near_sqrt
                int_division = near_rcp and (re.search(r"0x[4c]f800000", ins) or re.search(r"v_fmac_f32_e32 v\d+, 0, v\d+", ins))
                fmamk_2p32 = ins.startswith("v_fmamk_f32") and re.search(r"0x[4c]f800000", ins)   # (float)u64 = hi * 2^32 + lo
                if sqrt_residual or int_division or fmamk_2p32:
                    explained += 1
                else:
                    offenders.append((cur[:90], ins))
            recent.append(ins.split()[0])
    assert kernels >= 20, "expected the exact-path kernels in the code objects, found %d" % kernels
    assert not offenders, "fused multiply-add inside an exact-association kernel: %r" % offenders[:5]
    assert explained > 0   # the sqrt residuals are there: the scan did look inside the right functions


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="llvm-objdump not installed")
def test_streaming_kernels_address_lds_as_lds(tmp_path):
    so = tmp_path / "libmuopdb_hip.so"
    shutil.copy(L.LIB_PATH, so)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so.name], cwd=tmp_path, check=True, capture_output=True)
    flat, seen = {}, set()
    for f in [f for f in os.listdir(tmp_path) if f.endswith("gfx950")]:
        asm = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
        cur = None
        for line in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                seen.add(cur)
                continue
            if cur and line.split("//")[0].strip().startswith("flat_"):
                flat[cur] = flat.get(cur, 0) + 1
    checked = [k for k in seen if NO_FLAT.search(k)]
    assert len(checked) >= 20, "expected the streaming kernels in the code objects, found %d" % len(checked)
    bad = {k[:90]: v for k, v in flat.items() if NO_FLAT.search(k)}
    assert not bad, "flat (generic address space) memory instructions inside a streaming kernel: %r" % sorted(bad.items())[:5]
    beam = max((v for k, v in flat.items() if "hnsw_beam_kernel" in k), default=0)
    pipe = max((v for k, v in flat.items() if "hnsw_pipe_kernel" in k), default=0)
    assert pipe <= beam, "hnsw_pipe_kernel has %d flat instructions, the shared fallback accounts for %d" % (pipe, beam)
