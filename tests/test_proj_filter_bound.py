"""The error bound of the cull kernel's FILTERED PROJECTION (toyrenderer_amd/csrc/cull_math.hip.h: projBands, projectFiltered,
occTailQuadFiltered), checked on the CPU: tools/proj_filter_check.c replays the reference chain (culling.hlsli:53-78 under the
build's arithmetic convention) and the kernel's fast chain -- v_rsq_f32 / v_rcp_f32 modelled as ANY float within one ulp of
the correctly rounded value -- on random and adversarial spheres, and counts the lanes the kernel would call SURE whose level,
footprint origin or zero-weight flags differ from the reference's.  Must be none.  (The GPU side of the same claim:
tests/test_gpu_parity.py::test_projection_filter_at_its_decision_boundaries.)"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_filtered_projection_bound_holds_on_the_cpu_model(tmp_path):
    exe = str(tmp_path / "pfc")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-ffp-contract=off", os.path.join(ROOT, "tools", "proj_filter_check.c"), "-lm", "-o", exe])
    out = subprocess.run([exe, "20"], check=True, capture_output=True, text=True).stdout          # 20 M spheres, a few seconds on 8 cores
    assert re.search(r"mismatches among the sure lanes: 0\b", out), out
    sure = int(re.search(r"(\d+) sure", out).group(1))
    assert sure > 10_000_000, out
    worst = float(re.search(r"largest observed \|f - f'\| / bound: ([0-9.]+)", out).group(1))
    assert worst < 1.0, out                                                                           # the real-valued difference stays inside the proven bound
    # the bands are needed: without them the same spheres land at other levels / footprint origins
    m = re.search(r"without the bands: (\d+) spheres at another level, (\d+) at another footprint origin", out)
    assert int(m.group(1)) > 1000 and int(m.group(2)) > 100, out


def test_band_constants_are_the_same_in_the_kernel_and_in_the_cpu_model():
    """The CPU model restates cm::projBands; the constants of the bound must not drift apart."""
    k = open(os.path.join(ROOT, "toyrenderer_amd", "csrc", "cull_math.hip.h")).read()
    c = open(os.path.join(ROOT, "tools", "proj_filter_check.c")).read()
    for pat in (r"0\.51f \* \w*sqrtf\(B \* B \+ 1\.0f\) \+ 5\.11f \* B \+ 0\.15f \+ 12\.3f \* qmax",
                r"1\.125f / P\[i\] \+ 0\.25f", r"\(E1 \+ 12\.0f\) \* u", r"dim\[i\] \* \(E1 \+ 6\.0f\) \* u"):
        assert re.search(pat, k), pat
        assert re.search(pat, c), pat
    km = re.search(r"constexpr float kProjMargin = ([0-9.]+)f", k).group(1)
    cm = re.search(r"#define PROJ_MARGIN ([0-9.]+)f", c).group(1)
    assert km == cm
