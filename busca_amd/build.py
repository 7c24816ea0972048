"""Build libbusca_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "busca_hip.hip")
OUT = os.path.join(HERE, "libbusca_hip.so")


def _newest_source_mtime():
    m = 0.0
    for root in (os.path.join(HERE, "csrc"), os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build(force=False, verbose=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_source_mtime():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]
    if os.environ.get("BUSCA_CONV_PROBE"):      # s_memtime phase stamps in the ReID conv kernels (BUSCA_CONV_TS); costs ~1 %, off by default
        base.append("-DBUSCA_CONV_PROBE")
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs instead of AGPRs, which removes ~1 500 v_accvgpr_* copies from the
    # Decision-Transformer kernels (their epilogues are VALU work on the accumulators): f16 DT-step +7 %, f32 +1-2 %, ReID
    # unchanged (measured A/B on MI355X, round 2).  The pass behind it is young (it crashed on an experimental variant of the
    # kernel), so a failed compile falls back to the plain flags.
    variants = [["-mllvm", "-amdgpu-mfma-vgpr-form"], []] if os.environ.get("BUSCA_NO_VGPR_FORM") is None else [[]]
    r = None
    for extra in variants:
        cmd = base + extra + ["-o", OUT + ".tmp", SRC]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            break
        if extra:
            sys.stderr.write("hipcc failed with %s; retrying without it\n" % " ".join(extra))
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building libbusca_hip.so")
    os.replace(OUT + ".tmp", OUT)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
