"""Build libonet_hip.so in-tree with hipcc for gfx950 (MI355X).  No cmake, no torch extension:
the library is a plain C-ABI shared object (include/onet_hip.h) loaded with ctypes."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libonet_hip.so")
SOURCES = ["abi.cpp", "conv_mfma.hip", "conv_wino.hip", "conv_wino4.hip", "conv_wino4w.hip", "conv_bf16.hip", "bn.hip", "spatial.hip", "head_loss.hip", "optim.hip", "evalside.hip"]
ARCH = "gfx950"


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "onet_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.rsplit(".", 1)[0] + ".o")
        cmd = [hipcc, "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-fPIC", "-std=c++17", "-Wall",
               "-Wno-unused-result", "-c", os.path.join(CSRC, src), "-o", obj]
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
    subprocess.check_call([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
