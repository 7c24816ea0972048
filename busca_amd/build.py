"""Build libbusca_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

The library is several translation units (csrc/busca_internal.hpp lists them) compiled IN PARALLEL into busca_amd/build/*.o and linked: a full build is
as long as its slowest unit, and an edit recompiles only the units that include the edited file."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "libbusca_hip.so")
STAMP = OUT + ".flags"
# (unit, compiled with -amdgpu-mfma-vgpr-form where that compiles).  busca_dt_aux: the instantiations that crash that pass (see the file).
UNITS = [("busca_hip", True), ("busca_dt_f32", True), ("busca_dt_f16", True), ("busca_dt_x3", True), ("busca_dt_aux", False),
         ("busca_dtl_f32", True), ("busca_dtl_f16", True), ("busca_dtl_x3", True), ("busca_reid", True)]
VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]


def _requested_flags():
    """The flag set this environment asks for (BUSCA_CONV_PROBE, BUSCA_NO_VGPR_FORM change it)."""
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value"]
    if os.environ.get("BUSCA_CONV_PROBE"):      # s_memtime phase stamps in the ReID conv kernels (BUSCA_CONV_TS); costs ~1 %, off by default
        flags.append("-DBUSCA_CONV_PROBE")
    if os.environ.get("BUSCA_SPLIT_RELAXED"):   # A/B only: the token-split hand-off without its agent-scope fences (dt_kernel.hip.inc)
        flags.append("-DBUSCA_SPLIT_RELAXED")
    return flags


def _stamp_request():
    return " ".join(_requested_flags()) + (" [no-vgpr-form]" if os.environ.get("BUSCA_NO_VGPR_FORM") is not None else " [vgpr-form if it compiles]")


_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path, seen=None):
    """The unit's source and every quoted include below it (recursively)."""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for inc in _INC.findall(open(path, errors="replace").read()):
        _deps(os.path.join(os.path.dirname(path), inc), seen)
    return seen


def _newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def _compile(hipcc, unit, vgpr, used_line, verbose):
    """One unit -> build/<unit>.o.  -amdgpu-mfma-vgpr-form: MFMA results land in VGPRs instead of AGPRs, which removes ~1 500 v_accvgpr_* copies from the
    Decision-Transformer kernels (their epilogues are VALU work on the accumulators): f16 DT-step +7 %, f32 +1-2 %, ReID unchanged (measured A/B on
    MI355X, round 2).  The pass behind it is young (it crashed on an experimental variant of the kernel), so a unit that fails to compile with it is
    retried with the plain flags."""
    src, obj = os.path.join(CSRC, unit + ".hip"), os.path.join(OBJ, unit + ".o")
    base = [hipcc] + [f for f in _requested_flags() if f != "-shared"] + ["-c", "-o", obj + ".tmp", src, '-DBUSCA_BUILD_FLAGS="%s"' % used_line]
    tries = [VGPR_FORM, []] if vgpr else [[]]
    r = None
    for extra in tries:
        cmd = base + extra
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            os.replace(obj + ".tmp", obj)
            return unit, bool(extra), r.stderr
        if extra:
            sys.stderr.write("hipcc failed on %s with %s; retrying without it\n" % (unit, " ".join(extra)))
    raise RuntimeError("hipcc failed building %s.o:\n%s%s" % (unit, r.stdout, r.stderr))


def build(force=False, verbose=False):
    """(Re)build what is stale: a unit whose object is older than any file it includes, or everything when the library was built for another flag request
    (sidecar stamp libbusca_hip.so.flags: line 1 = the request, line 2 = the flags that compiled - busca_build_info() returns line 2).  Without hipcc
    (a GPU box that received the built library) an existing library is used as it is."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    have_hipcc = os.path.exists(hipcc)
    stamp_ok = os.path.exists(STAMP) and open(STAMP).read().split("\n")[0] == _stamp_request()
    all_deps = set()
    for unit, _ in UNITS:
        all_deps |= _deps(os.path.join(CSRC, unit + ".hip"))
    all_deps.add(os.path.join(os.path.dirname(HERE), "include", "busca_hip.h"))
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= _newest(all_deps) and (stamp_ok or not have_hipcc):
        return OUT
    if not have_hipcc:
        raise RuntimeError("%s is stale or missing and hipcc (%s) is not here" % (OUT, hipcc))
    os.makedirs(OBJ, exist_ok=True)
    want_vgpr = os.environ.get("BUSCA_NO_VGPR_FORM") is None
    used_line = " ".join(_requested_flags() + (VGPR_FORM if want_vgpr else []))
    stale = []
    for unit, vgpr in UNITS:
        obj = os.path.join(OBJ, unit + ".o")
        if force or not stamp_ok or not os.path.exists(obj) or os.path.getmtime(obj) < _newest(_deps(os.path.join(CSRC, unit + ".hip"))):
            stale.append((unit, vgpr and want_vgpr))
    jobs = int(os.environ.get("BUSCA_BUILD_JOBS", "0")) or min(len(stale) or 1, os.cpu_count() or 4)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        for unit, with_vgpr, warn in ex.map(lambda uv: _compile(hipcc, uv[0], uv[1], used_line, verbose), stale):
            if verbose and warn.strip():
                sys.stderr.write(warn)
    # the objects carry their device code already (no relocatable device code): a plain host link against the HIP runtime
    clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang++")
    clang = clang if os.path.exists(clang) else "/opt/rocm/lib/llvm/bin/clang++"
    cmd = [clang, "-shared", "-fPIC", "--hip-link", "--offload-arch=gfx950", "-o", OUT + ".tmp"] + [os.path.join(OBJ, u + ".o") for u, _ in UNITS]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("hipcc failed linking libbusca_hip.so")
    os.replace(OUT + ".tmp", OUT)
    with open(STAMP, "w") as f:
        f.write(_stamp_request() + "\n" + used_line + "\n")
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
