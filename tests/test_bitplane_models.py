"""The numpy models of the bit-plane kernels (profiles/bitplane_model.py: k_pack_bits + k_local_bits; profiles/bitplane_coarse_model.py: the
coarse pass of make CBITS=1) against the byte evaluation and the oracle, on a sample of the bench workload.  CPU only: they state the
layouts and the lane-level algorithms the HIP kernels implement."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, step):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", script), str(step)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def test_refinement_model_equals_byte_evaluation_and_oracle():
    text = _run("bitplane_model.py", 100)
    assert "bit-plane == byte evaluation: True" in text, text
    assert "equal (as multisets): True" in text, text


def test_coarse_model_equals_byte_evaluation():
    text = _run("bitplane_coarse_model.py", 50)
    assert "bit-plane coarse pass == byte evaluation: True" in text, text


def test_frontend_bitsliced_vote_and_median_equal_the_oracle():
    """The carry-save window counts of the front end (3x3 majority vote, 5x5 median of one-hot normals) as a numpy model against the oracle."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "profiles", "frontend_bitslice_model.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr[-2000:]
    assert "bit-sliced vote == oracle hysteresis_gradient: True" in out.stdout
    assert "bit-sliced median == median filter: True" in out.stdout
    assert "half records by 8x8 bit transposes == direct packing: True" in out.stdout
