"""Build libctgcn_hip.so for gfx950 in-tree:  python -m ctgcn_amd.build [--force]"""
import hashlib
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
SRCS = [os.path.join(_HERE, "csrc", f) for f in ("ctgcn_hip.hip", "ctgcn_gemm.hip", "ctgcn_gru_bwd.hip", "ctgcn_ingest.hip", "ctgcn_walks.hip", "ctgcn_export.cpp")]
HDR = os.path.join(os.path.dirname(_HERE), "include", "ctgcn_hip.h")
JITTER_HDR = os.path.join(_HERE, "csrc", "ctgcn_jitter.h")     # included by the kernel files (inert without -DCTGCN_JITTER): part of the source hash
TABLE_HDR = os.path.join(_HERE, "csrc", "ctgcn_table.h")       # descriptor-table upload of the grouped launches (both kernel files)
OUT = os.path.join(_HERE, "csrc", "libctgcn_hip.so")
STAMP = OUT + ".srchash"          # sha256 of the sources + header + this recipe the .so was built from (travels with it, git-ignored)


def source_hash():
    h = hashlib.sha256()
    for f in SRCS + [HDR, JITTER_HDR, TABLE_HDR, os.path.abspath(__file__)]:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def is_current():
    """True when libctgcn_hip.so exists and was built from exactly the sources in the tree (content hash, not mtimes: a snapshot copied
    to another box gets fresh mtimes in arbitrary order, and a stale-but-newer binary must not be trusted)."""
    if not (os.path.exists(OUT) and os.path.exists(STAMP)):
        return False
    with open(STAMP) as fh:
        return fh.read().strip() == source_hash()


def build(force=False, verbose=False):
    if not force and is_current():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -amdgpu-mfma-vgpr-form (ctgcn_hip.hip only): MFMA accumulators in architectural VGPRs.  gru_layer_h2_kernel fills the AGPR half
    # of the register file with pinned weight fragments; with the default (AGPR-form) accumulators every result crosses back through
    # v_accvgpr_read before the gate math (24 % of that kernel's VALU instructions).  ctgcn_gemm.hip wants the opposite: the
    # 256 x 128 GEMM tile keeps its 128 accumulator registers in the AGPR half and 246 operand / staging registers in the VGPR half.
    common = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-pthread"]
    if verbose:
        common.append("-Rpass-analysis=kernel-resource-usage")
    objs, procs = [], []
    try:
        for src in SRCS:
            obj = os.path.splitext(src)[0] + ".o"
            extra = ["-mllvm", "-amdgpu-mfma-vgpr-form"] if os.path.basename(src) == "ctgcn_hip.hip" else []
            objs.append(obj)
            procs.append((src, subprocess.Popen(common + extra + ["-c", src, "-o", obj])))
        failed = [(src, pr.returncode) for src, pr in procs if pr.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(failed[0][1], "hipcc -c " + failed[0][0])
        if os.path.exists(STAMP):
            os.remove(STAMP)
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", OUT] + objs)
        with open(STAMP, "w") as fh:
            fh.write(source_hash() + "\n")
    finally:
        for _, pr in procs:         # a failed spawn leaves the earlier compiles running: do not orphan them
            if pr.poll() is None:
                pr.kill()
                pr.wait()
        for obj in objs:
            if os.path.exists(obj):
                os.remove(obj)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
