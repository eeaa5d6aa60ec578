"""Build libctgcn_hip.so for gfx950 in-tree:  python -m ctgcn_amd.build [--force]"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(_HERE, "csrc", f) for f in ("ctgcn_hip.hip", "ctgcn_gemm.hip", "ctgcn_ingest.hip", "ctgcn_walks.hip", "ctgcn_export.cpp")]
HDR = os.path.join(os.path.dirname(_HERE), "include", "ctgcn_hip.h")
OUT = os.path.join(_HERE, "csrc", "libctgcn_hip.so")


def build(force=False, verbose=False):
    newest = max([os.path.getmtime(f) for f in SRCS] + [os.path.getmtime(HDR)])
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= newest:
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form: MFMA accumulators in architectural VGPRs.  gru_layer_h2_kernel fills the AGPR half of the register
    # file with pinned weight fragments; with the default (AGPR-form) accumulators every result crosses back through
    # v_accvgpr_read before the gate math (24 % of that kernel's VALU instructions).  The other kernels use no AGPRs either way.
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-mllvm", "-amdgpu-mfma-vgpr-form", "-shared", "-fPIC", "-pthread",
           "-o", OUT] + SRCS
    if verbose:
        cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
