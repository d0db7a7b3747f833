"""The kernels that were measured and lost live in experiments builds only (make EXTRA=-DLDM_EXPERIMENTS OUT=../libldm3d_exp.so, built by
__graft_entry__.build() next to the product library).  DESIGN.md quotes them as "parity-green, slower": this file keeps that first half
true.  Every case runs the operator parity tests of tests/test_gpu_ops.py in ONE child process with the experiments library and the
kernel's switch in its environment (the switches are read once per process), one child at a time."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "3d-latent-diffusion-model_amd", "libldm3d_exp.so")

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("knob,value,select", [
    ("LDM_HALO_RW", "1", "conv and not wgrad and not dgrad"),      # conv3_halo_rw_kernel: register-fed weights, one barrier per macro step
    ("LDM_HALO_PP", "1", "conv and not wgrad and not dgrad"),      # conv3_halo_pp_kernel: alternating K steps per wave group, one barrier per six steps
    ("LDM_WGRAD_KW3", "1", "wgrad"),                               # conv_wgrad_kw3_kernel: three kw taps per workgroup
    ("LDM_WGRAD_KW3", "2", "wgrad"),                               # ... its sixteen-wave form
])
def test_lost_kernels_stay_parity_green(cuda, knob, value, select):
    if not os.path.exists(EXP):
        pytest.skip("no experiments build (libldm3d_exp.so)")
    env = dict(os.environ, LDM3D_LIB=EXP)
    env[knob] = value
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_ops.py"), "-x", "-q", "-k", select,
                        "-p", "no:cacheprovider"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    tail = (r.stdout or "")[-1500:] + (r.stderr or "")[-500:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout, tail
