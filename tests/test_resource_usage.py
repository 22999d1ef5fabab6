"""Resource-usage regression gate (round 3 review, item 7): every kernel of the default paths compiles to ScratchSize 0 with no VGPR
spill and inside the register cap of its workgroup size.  Compiles both translation units with the Makefile's flags and
`-Rpass-analysis=kernel-resource-usage` (tools/resource_table.py; ~70 s, no GPU needed).  The table of this commit is
profiles/r4_resource_usage.txt."""
import os
import re
import sys

import pytest

from conftest import REPO

sys.path.insert(0, os.path.join(REPO, "tools"))

# kernels a default run launches at some batch size: (name pattern, VGPR cap of its workgroup size: 512 / waves per SIMD)
DEFAULT_VIT48 = [
    # large batches (what bench.py times)
    (r"vts::stem_fused_kernel<[012], false, true>", 128),            # 1024 threads
    (r"vts::stem_stream_kernel<256, 128, [012]>", 128),
    (r"vtb::blocks_kernel<5, 8, 1, true, true, (false|true), true>", 256),       # 512 threads
    (r"vtb::blocks_kernel<20, 8, 3, false, false, (false|true), true>", 256),
    (r"vth3::head_fused3_kernel", 168),                              # 768 threads
    (r"vth3::head_seq3_kernel<8, 2, false>", 256),
    # small batches / the plugin's one-sequence step
    (r"vts::stem_a_kernel", 256), (r"vts::stem_b_kernel<false>", 256),
    (r"vtb::tile_qkv_kernel<(5|20)>", 512), (r"vtb::tile_attn_mlp_kernel<(5|20)>", 256),
    (r"vth3::head_towers3_kernel", 256), (r"vth::head_towers_kernel<16, 8, false, (false|true)>", 256), (r"vth::head_conv1_kernel<16>", 256),
    (r"vth::decode_kernel", 512), (r"vtt::crop_kernel<(false|true)>", 256), (r"vtt::update_state_kernel", 512),
    # fp32-MFMA forms selected by VT_*_BF3=0 (bench.py's all-fp32 comparison)
    (r"vts::stem_fused_kernel<[012], false, false>", 128), (r"vtb::blocks_kernel<5, 8, 1, true, true, (false|true), false>", 256),
    (r"vtb::blocks_kernel<20, 8, 3, false, false, (false|true), false>", 256), (r"vth::head_fused_kernel<8, false>", 168),
    (r"vth::head_seq_kernel<16, 8, false>", 256),
]
# vts::stem_a2_kernel reports ScratchSize 36 with VGPRs Spill 0 and not one scratch instruction: SGPRs spilled to VGPR lanes reserve a
# frame that is never touched.  It must stay free of VGPR spills.
SGPR_FRAME_ONLY = [r"vts::stem_a2_kernel"]
# ViT-Base: four of the six 256 x 256 GEMM instantiations park one or two finished accumulator quads in scratch in a tile's LAST
# k-tile and reload them in the epilogue -- once per tile, outside the k-loop.  Bounded here so it cannot grow; the
# k-loop instantiation without spills (GELU epilogue) and everything else must stay at zero.
VITB_BOUNDED = {r"vbg::gemm_kernel<256, 256, 2, 4, 0, (0|1|2|5)>": (16, 36), r"vbg::gemm_kernel<256, 256, 2, 4, 1, 4>": (4, 20)}


@pytest.fixture(scope="module")
def tables():
    import resource_table as rt
    return {"vittrack": rt.table("vittrack.hip"), "vitb": rt.table("vitb.hip")}


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
        assert r["vspill"] == 0, (r["name"], r)
        if r["scratch"]:
            assert any(re.fullmatch(p, r["name"]) for p in SGPR_FRAME_ONLY), (r["name"], r)


def test_vitb_kernels(tables):
    for r in tables["vitb"]:
        bound = next((b for p, b in VITB_BOUNDED.items() if re.fullmatch(p, r["name"])), None)
        if bound is None:
            assert r["scratch"] == 0 and r["vspill"] == 0, (r["name"], r)
        else:
            assert r["vspill"] <= bound[0] and r["scratch"] <= bound[1], (r["name"], r, bound)


def test_committed_table_is_current(tables):
    """profiles/r4_resource_usage.txt is this commit's table (regenerate: python tools/resource_table.py > profiles/r4_resource_usage.txt)."""
    txt = open(os.path.join(REPO, "profiles", "r4_resource_usage.txt")).read()
    for key in ("vittrack", "vitb"):
        for r in tables[key]:
            if not r["name"]:
                continue
            line = next((ln for ln in txt.splitlines() if ln.startswith(r["name"][:96].ljust(96))), None)
            assert line is not None, r["name"]
            assert f"vgpr {r['vgpr']:4d}  scratch {r['scratch']:4d}" in line, (r["name"], line, r)
