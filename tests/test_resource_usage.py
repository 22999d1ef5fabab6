"""Resource-usage regression gate (round 3 review, item 7): every kernel of the default paths compiles to ScratchSize 0 with no VGPR
spill and inside the register cap of its workgroup size.  Compiles both translation units with the Makefile's flags and
`-Rpass-analysis=kernel-resource-usage` (tools/resource_table.py; ~2 min, no GPU needed; skipped where hipcc is absent).  The table
of the round's last kernel change is profiles/r6_resource_usage.txt: this commit may not use MORE registers or scratch than it says."""
import os
import re
import shutil
import sys

import pytest

from conftest import REPO

if shutil.which("hipcc") is None:
    pytest.skip("hipcc not on PATH: the resource gate needs the compiler", allow_module_level=True)

sys.path.insert(0, os.path.join(REPO, "tools"))

# kernels a default run launches at some batch size: (name pattern, VGPR cap of its workgroup size: 512 / waves per SIMD)
DEFAULT_VIT48 = [
    # large batches (what bench.py times)
    (r"vts::stem_fused_kernel<[012], false, true, false>", 128),     # 1024 threads
    (r"vts::stem_fused_kernel<1, false, (false|true), true>", 128),  # round 6: the tracker step's uint8-patch forms
    (r"vts::stem_stream_kernel<256, 128, [012], false>", 128), (r"vts::stem_stream_kernel<(256, 128|128, 64), 1, true>", 128),
    (r"vtb::blocks_kernel<5, 8, 1, true, true, (false|true), true, (false|true)>", 256),       # 512 threads; last flag: A3 (round 5)
    (r"vtb::blocks_kernel<20, 8, 3, false, false, (false|true), true, false>", 256),       # VT_BLOCKS_BF3=1
    (r"vth3::head_fused3_kernel", 168),                              # 768 threads
    (r"vth3::head_seq3_kernel<8, 2, false>", 256),
    # small batches / the plugin's one-sequence step
    (r"vts::stem_a_kernel<(false|true)>", 256), (r"vts::stem_b_kernel<false>", 256),
    (r"vtb::tile_qkv_kernel<(5|20)>", 512), (r"vtb::tile_attn_mlp_kernel<(5|20)>", 256),
    (r"vth3::head_towers3_kernel", 256), (r"vth::head_towers_kernel<16, 8, false, (false|true)>", 256), (r"vth::head_conv1_kernel<16>", 256),
    (r"vth::decode_kernel", 512), (r"vtt::crop_kernel<(false|true), (false|true)>", 256), (r"vtt::crop_fast_kernel<[124], (false|true)>", 256),
    (r"vtt::crop_band_kernel<(false|true), [456], [24], (false|true)>", 256), (r"vtt::update_state_kernel", 512),
    # fp32-MFMA forms selected by VT_*_BF3=0 (bench.py's all-fp32 comparison)
    (r"vts::stem_fused_kernel<[012], false, false, false>", 128), (r"vtb::blocks_kernel<5, 8, 1, true, true, (false|true), false, false>", 256),
    (r"vtb::blocks_kernel<20, 8, 3, false, false, (false|true), false, false>", 256), (r"vth::head_fused_kernel<8, false>", 168),
    (r"vth::head_seq_kernel<16, 8, false>", 256),
]
# vts::stem_a2_kernel reports ScratchSize 36 with VGPRs Spill 0 and not one scratch instruction: SGPRs spilled to VGPR lanes reserve a
# frame that is never touched.  It must stay free of VGPR spills.
SGPR_FRAME_ONLY = [r"vts::stem_a2_kernel",
                   # the shape-generic attention at a head dimension that is none of 16 / 32 / 48 / 64: its q / o arrays of run-time length
                   # live in scratch BY DESIGN (vt_generic.h: reference-implementation speed); no register is spilled
                   r"vtg::attn_kernel<0>"]
# The G256 block kernel with K as pieces (VT_BLOCKS_BF3=2, the default; round 5) parks the q of a wave's second and third token tile
# (6 float4) and one residual chunk in scratch across the qkv barrier and reloads each once where that tile's attention starts --
# seven 16-byte stores + loads per block and wave, outside every loop (the 20-tile form holds three tiles' residual streams and q
# at the 256-register cap; 136 B since P.V moved to the bf16 pipe: one more float4).  Bounded so it cannot grow.
PARKED = {r"vtb::blocks_kernel<20, 8, 3, false, false, (false|true), true, true>": (72, 136),      # (VGPRs spilled incl. SGPR-spill lanes, scratch bytes)
          # the DIAGNOSTIC instantiation (VT_DBG_STAMPS: tools/head_stamps.py) of a kernel whose production form sits at the 256-register cap
          # with 0 B: its stamp counter and buffer address are what spills
          r"vth3::head_seq3_kernel<8, 2, true>": (8, 32)}
# the f16 build (BASELINE config 5, -DVT_F16=1): the kernels its default path launches at B = 256 (round 4 advisor: they were printed
# in the table but never gated)
DEFAULT_F16 = [
    (r"vts::stem_fused_kernel<[012], false, (false|true), (false|true)>", 128), (r"vts::stem_pipe_kernel.*", 256), (r"vts::stem_b_kernel<false>", 256),
    (r"vtb::blocks_kernel<5, 8, 1, true, true, (false|true), false, false>", 256),
    (r"vtb::blocks_kernel<20, 8, 3, false, false, (false|true), false, false>", 256),
    (r"vth::head_fused_kernel<8, false>", 168), (r"vth::head_seq_kernel<16, 8, false>", 256),
]


@pytest.fixture(scope="module")
def tables():
    import resource_table as rt
    return {"vittrack": rt.table("vittrack.hip"), "vittrack_f16": rt.table("vittrack.hip", ("-DVT_F16=1",)), "vitb": rt.table("vitb.hip")}


def _find(rows, pat):
    hit = [r for r in rows if re.fullmatch(pat, r["name"])]
    assert hit, f"no kernel matches {pat!r}: the list in this test is stale"
    return hit


def test_default_vit48_kernels_have_no_scratch_and_fit_their_register_cap(tables):
    rows = tables["vittrack"]
    for pat, cap in DEFAULT_VIT48:
        for r in _find(rows, pat):
            assert r["scratch"] == 0 and r["vspill"] == 0, (r["name"], r)
            assert r["vgpr"] <= cap, (r["name"], r["vgpr"], cap)


def test_no_vit48_kernel_spills_vector_registers(tables):
    for r in tables["vittrack"]:
        bound = next((b for p, b in PARKED.items() if re.fullmatch(p, r["name"])), None)
        if bound is not None:
            assert r["vspill"] <= bound[0] and r["scratch"] <= bound[1], (r["name"], r, bound)
            continue
        assert r["vspill"] == 0, (r["name"], r)
        if r["scratch"]:
            assert any(re.fullmatch(p, r["name"]) for p in SGPR_FRAME_ONLY), (r["name"], r)


def test_default_f16_kernels_have_no_scratch_and_fit_their_register_cap(tables):
    rows = tables["vittrack_f16"]
    for pat, cap in DEFAULT_F16:
        for r in _find(rows, pat):
            assert r["scratch"] == 0 and r["vspill"] == 0, (r["name"], r)
            assert r["vgpr"] <= cap, (r["name"], r["vgpr"], cap)


def test_vitb_kernels(tables):
    """Round 5: every ViT-Base kernel at ScratchSize 0 (the GEMMs' bias left the registers: vb_gemm.h)."""
    for r in tables["vitb"]:
        assert r["scratch"] == 0 and r["vspill"] == 0, (r["name"], r)


def test_committed_table_is_not_exceeded(tables):
    """profiles/r6_resource_usage.txt (python tools/resource_table.py > profiles/r6_resource_usage.txt) is a ceiling: a kernel of this
    commit may use fewer registers / less scratch than the table says (another hipcc, a later edit), never more."""
    heads = {"## vittrack.hip": "vittrack", "## vittrack.hip -DVT_F16=1": "vittrack_f16", "## vitb.hip": "vitb"}
    committed, cur = {k: {} for k in heads.values()}, None
    for ln in open(os.path.join(REPO, "profiles", "r6_resource_usage.txt")):
        if ln.startswith("## "):
            cur = heads[ln.strip()]
            continue
        m = re.search(r"^(.{96}) vgpr\s+(\d+)\s+scratch\s+(\d+)", ln)
        if m and cur:
            committed[cur][m.group(1).rstrip()] = (int(m.group(2)), int(m.group(3)))
    for key, rows in tables.items():
        for r in rows:
            if not r["name"]:
                continue
            assert r["name"][:96] in committed[key], (key, r["name"])
            vg, sc = committed[key][r["name"][:96]]
            assert r["vgpr"] <= vg and r["scratch"] <= sc, (key, r["name"], (vg, sc), r)
