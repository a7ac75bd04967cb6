"""Register budget of the kernels whose occupancy was measured to matter (no GPU needed: read
from the code object's metadata in the built library).  dc_eval runs 512-thread workgroups;
above 128 VGPRs only one fits a CU, which costs ~20 % when 8-32 chains share a GPU
(DESIGN.md section 5), and scratch means spilling in the evaluation's hot loop."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bpl-next_amd", "bpl", "libbplhip.so")


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    tools = [os.path.join(LLVM, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not os.path.exists(LIB) or not all(os.path.exists(t) for t in tools):
        pytest.skip("library or LLVM tools not present")
    d = tmp_path_factory.mktemp("co")
    fat, co = str(d / "fat.bin"), str(d / "gfx950.co")
    subprocess.run([tools[0], "--dump-section", f".hip_fatbin={fat}", LIB], check=True)
    subprocess.run([tools[1], "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
    out = {}
    for block in notes.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        out[name] = {
            "vgpr": int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1)),
            "scratch": int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1)),
        }
    shutil.rmtree(d, ignore_errors=True)
    return out


def _eval_name(weighted, clip, staged, nuts):
    b = lambda v: "Lb1E" if v else "Lb0E"
    # (round 4: five pointers and four ints in front of the argument block -- the ones gfx950 preloads into SGPRs)
    return f"_ZN2dc7dc_evalI{b(weighted)}{b(clip)}{b(staged)}{b(nuts)}EEvPKdPKjS4_S4_S4_iiiiNS_8EvalArgsE"


def test_every_eval_variant_is_built(kernels):
    for w in (0, 1):
        for c in (0, 1):
            for s in (0, 1):
                for n in (0, 1):
                    assert _eval_name(w, c, s, n) in kernels


def test_headline_kernels_fit_two_workgroups_per_cu(kernels):
    # the basic model, <= 64 teams: plain evaluation and the NUTS-aware launch
    for nuts in (0, 1):
        k = kernels[_eval_name(0, 0, 1, nuts)]
        assert k["vgpr"] <= 128, k
        assert k["scratch"] == 0, k
    # the extended model's NUTS-aware launch stays at two workgroups per CU as well
    k = kernels[_eval_name(0, 1, 1, 1)]
    assert k["vgpr"] <= 128 and k["scratch"] == 0, k


def test_no_spills_in_unweighted_staged_kernels(kernels):
    for c in (0, 1):
        for n in (0, 1):
            assert kernels[_eval_name(0, c, 1, n)]["scratch"] == 0
