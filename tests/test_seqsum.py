"""jlama_amd/csrc/jh_seqsum.h -- the integer form of a float running sum that sample_pick_kernel evaluates with 1024 lanes --
checked on the host against the plain loop of AbstractModel.sample (AbstractModel.java:475-489): tests/native/seqsum_harness.hip
is compiled with hipcc (its main() runs on the CPU, no HIP call) and must report no mismatch."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_parallel_float_running_sum_is_the_sequential_one(tmp_path):
    exe = str(tmp_path / "seqsum_harness")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off",
                    os.path.join(HERE, "native", "seqsum_harness.hip"), "-o", exe], check=True, capture_output=True, text=True)
    out = subprocess.run([exe], check=False, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:]
    assert "failures 0" in out.stdout
