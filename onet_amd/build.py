"""Build libonet_hip.so in-tree with hipcc for gfx950 (MI355X).  No cmake, no torch extension:
the library is a plain C-ABI shared object (include/onet_hip.h) loaded with ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libonet_hip.so")
SOURCES = ["abi.cpp", "conv_mfma.hip", "conv_wino4.hip", "conv_wino4w.hip", "conv_split.hip", "convt_gemm.hip", "stem.hip", "bn.hip", "spatial.hip", "head_loss.hip", "optim.hip", "evalside.hip", "clutter.hip"]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp", ".hpp"))] + \
        [os.path.join(HERE, "..", "include", "onet_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, variant=None, flags=()):
    """variant / flags: an experimental build for same-box A/B runs -> libonet_hip_<variant>.so (objects under
    csrc/_<variant>/), selected at run time with ONET_HIP_LIB=<path>; the default library is untouched."""
    lib = LIB if variant is None else os.path.join(HERE, f"libonet_hip_{variant}.so")
    if variant is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = CSRC if variant is None else os.path.join(CSRC, "_" + variant)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.rsplit(".", 1)[0] + ".o")
        # -fno-slp-vectorize: hipcc's SLP pass packs adjacent scalar fp32 adds / fmas of the Winograd transforms into
        # v_pk_add_f32 / v_pk_fma_f32, which cost 11-13 cycles more than the pair of scalar ops beside MFMAs
        # (MI355X_MICROARCH.md, 'packed f32 VALU ... an anti-lever'); same-box: F(4x4) fwd +1.5 %, F(3x3,4x4) wgrad +4 %
        cmd = [hipcc, "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall",
               "-Wno-unused-result", "-fno-slp-vectorize", *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            cmd.insert(-4, "-Rpass-analysis=kernel-resource-usage")
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[onet_amd.build] {src} failed:\n{out}\n")
        elif verbose or out.strip():
            sys.stderr.write(out)
    if failed:
        raise RuntimeError("hipcc failed building libonet_hip.so")
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:        # python -m onet_amd.build --variant NAME -DFLAG=1 -fno-slp-vectorize ...
        i = sys.argv.index("--variant")
        print(build(variant=sys.argv[i + 1], flags=[a for a in sys.argv[i + 2:]]))
    else:
        print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
