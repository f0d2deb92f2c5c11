"""CPU: the register prefetch of the bf16 / fp32 convolution kernels must stay un-exposed in the generated code.

The kernels fetch chunk k+1 into registers and only then run the MFMA section of chunk k.  Twice the compiler has put a
`s_waitcnt vmcnt(0)` between the two (a run-time choice between two weight-fetch variants; selects on freshly loaded values),
which makes every chunk wait for its own prefetch - results stay correct, the kernel just loses its overlap (DESIGN.md
section 5).  This test compiles the two sources to gfx950 assembly (hipcc cross-compiles without a GPU) and checks the order
of events in every main loop with tools/isa_events.py: loads, then the MFMA section, no vector-memory wait in between."""
import os
import shutil
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "continual-learning-for-dynamic-video-quality-enhancement_amd", "csrc")
sys.path.insert(0, os.path.join(REPO, "tools"))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _asm(tmp_path, name):
    out = tmp_path / (name + ".s")
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-S", "--cuda-device-only",
                        "-I" + os.path.join(REPO, "include"), os.path.join(CSRC, name + ".hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return str(out)


def _main_loops(events, min_mfma):
    """(index of the MFMA section, events since the previous barrier) for every MFMA run of at least min_mfma instructions"""
    for i, e in enumerate(events):
        if e.startswith("mfma x") and int(e[6:]) >= min_mfma:
            j = i
            while j > 0 and events[j - 1] != "BARRIER":
                j -= 1
            yield i, [x for x in events[j:i] if not (x.startswith(".LBB") or x.startswith("s_c") or x.startswith("s_branch"))]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("source,pattern,min_mfma", [
    ("conv_bf16", "conv_bf16_kernelILi2ELi3ELb1ELi8ELi1", 72),      # dense-layer 3x3 conv, 16x32 tiles (the dominant kernel)
    ("conv_bf16", "conv_bf16_kernelILi2ELi3ELb1ELi8ELi2", 72),      # 64-channel 3x3 conv, channel-split
    ("conv_bf16", "conv_bf16_kernelILi2ELi3ELb1ELi4ELi1", 72),
    ("conv_bf16", "conv_bf16_kernelILi4ELi3ELb0ELi4ELi1", 144),     # fp32-stored input
    ("conv_bf16", "rdb_tail_kernel", 80),
    ("conv_bf16", "rdb_tail8_kernel", 72),                         # the 3x3 waves' loop of the two-role tail kernel
    ("conv_igemm", "conv_f32_kernelILi2ELi3ELi1", 288),             # exact-fp32 mode
])
def test_prefetch_is_not_waited_for_before_the_mfma_section(tmp_path, source, pattern, min_mfma):
    import isa_events
    cache = os.path.join(str(tmp_path.parent), source + ".s")       # one compile per source and test session
    if not os.path.exists(cache):
        shutil.copy(_asm(tmp_path, source), cache)
    found = 0
    for name, body in isa_events.kernels(cache):
        if pattern not in name:
            continue
        for _, before in _main_loops(isa_events.events(body), min_mfma):
            loads = [k for k, e in enumerate(before) if e.startswith("LOAD")]
            if not loads:
                continue                                            # (an MFMA section that follows no fetch: the last chunk)
            found += 1
            after_last_load = before[loads[-1] + 1:]
            assert not any("vmcnt" in e for e in after_last_load), (name, before)
            assert not any(e == "SCRATCH!" for e in before), (name, "scratch access in the main loop")
    assert found >= 1, f"no main loop found for {pattern}"
