"""Build libbusca_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "busca_hip.hip")
AUX = os.path.join(HERE, "csrc", "busca_dt_aux.hip")      # instantiations that must be compiled without -amdgpu-mfma-vgpr-form (see the file)
OUT = os.path.join(HERE, "libbusca_hip.so")


def _newest_source_mtime():
    m = 0.0
    for root in (os.path.join(HERE, "csrc"), os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


STAMP = OUT + ".flags"


def _requested_flags():
    """The flag set this environment asks for (BUSCA_CONV_PROBE, BUSCA_NO_VGPR_FORM change it)."""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]
    if os.environ.get("BUSCA_CONV_PROBE"):      # s_memtime phase stamps in the ReID conv kernels (BUSCA_CONV_TS); costs ~1 %, off by default
        flags.append("-DBUSCA_CONV_PROBE")
    return flags


def _stamp_request():
    return " ".join(_requested_flags()) + (" [no-vgpr-form]" if os.environ.get("BUSCA_NO_VGPR_FORM") is not None else " [vgpr-form if it compiles]")


def build(force=False, verbose=False):
    """(Re)build when a source is newer than the library OR the library was built for another flag request (sidecar stamp
    libbusca_hip.so.flags: line 1 = the request, line 2 = the flags that actually compiled - busca_build_info() returns line 2)."""
    stamp_ok = os.path.exists(STAMP) and open(STAMP).read().split("\n")[0] == _stamp_request()
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest_source_mtime() and (stamp_ok or not os.path.exists("/opt/rocm/bin/hipcc")):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc] + _requested_flags()
    # -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs instead of AGPRs, which removes ~1 500 v_accvgpr_* copies from the
    # Decision-Transformer kernels (their epilogues are VALU work on the accumulators): f16 DT-step +7 %, f32 +1-2 %, ReID
    # unchanged (measured A/B on MI355X, round 2).  The pass behind it is young (it crashed on an experimental variant of the
    # kernel), so a failed compile falls back to the plain flags.
    variants = [["-mllvm", "-amdgpu-mfma-vgpr-form"], []] if os.environ.get("BUSCA_NO_VGPR_FORM") is None else [[]]
    aux_obj = OUT + ".aux.o"
    cmd = [hipcc] + [f for f in _requested_flags() if f != "-shared"] + ["-c", "-o", aux_obj, AUX]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed building busca_dt_aux.o")
    used = None
    for extra in variants:
        used = " ".join(base[1:] + extra)
        cmd = base + extra + ['-DBUSCA_BUILD_FLAGS="%s"' % used, "-o", OUT + ".tmp", SRC, "-Wl," + aux_obj]      # (a bare .o after a .hip source is parsed as HIP source)
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
    os.remove(aux_obj)
    with open(STAMP, "w") as f:
        f.write(_stamp_request() + "\n" + used + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
