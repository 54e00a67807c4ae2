"""Build the gfx950 shared objects in-tree with hipcc (cross-compiles without a GPU).

    python -m mila_amd.build            # incremental
    python -m mila_amd.build --force

Outputs (git-ignored, but they travel with gpurun):
    mila_amd/lib/libmila_cdna4.so       the C-ABI device backend (include/mila_cdna4.h)
    mila_amd/lib/libmila_host.so        the C++ host mirror's C entry points (model runners)
    mila_amd/lib/libmila_cdna4_experiments.so   csrc/experiments/: the measured-slower in-launch decode forms (chain, engine); tests / tools only
"""
import concurrent.futures as cf
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "csrc")
HOST = os.path.join(ROOT, "host")
LIBDIR = os.path.join(ROOT, "lib")
OBJDIR = os.path.join(ROOT, "lib", "obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

KERNEL_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                "-Wall", "-Wno-unused-function"]
# host mirror: plain C++23 (no device code), HIP runtime API only for graphs/events
HOST_FLAGS = ["-O2", "-std=c++23", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
              "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(os.path.dirname(ROOT), "include"),
              "-I" + os.path.join(HOST, "include")]
HOSTCXX = os.environ.get("HOSTCXX", "/opt/rocm/lib/llvm/bin/clang++")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers(d):
    out = []
    for r, _, fs in os.walk(d):
        out += [os.path.join(r, f) for f in fs if f.endswith((".h", ".hpp"))]
    return out


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return p.stdout


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    inc = os.path.join(os.path.dirname(ROOT), "include")
    hdrs = _headers(CSRC) + _headers(inc)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + KERNEL_FLAGS + ["-c", src, "-o", obj])
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    lib = os.path.join(LIBDIR, "libmila_cdna4.so")
    if force or jobs or _newer(lib, objs):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs)
        # a kernel whose host stub the compiler dropped (seen with lambdas capturing arrays inside a kernel template: no diagnostic) only fails at dlopen
        missing = [ln for ln in _run(["nm", "-u", lib]).splitlines() if "__device_stub__" in ln]
        if missing:
            os.remove(lib)
            raise RuntimeError("kernels without a host stub in %s:\n%s" % (lib, "\n".join(missing)))

    # experiments (csrc/experiments/*.hip): a library of their own, linked against the product's, loaded only by tests/ and tools/
    exp_dir = os.path.join(CSRC, "experiments")
    if os.path.isdir(exp_dir):
        eobjs, ejobs = [], []
        for s_ in sorted(f for f in os.listdir(exp_dir) if f.endswith(".hip")):
            src = os.path.join(exp_dir, s_)
            obj = os.path.join(OBJDIR, "exp_" + s_[:-4] + ".o")
            eobjs.append(obj)
            if force or _newer(obj, [src] + hdrs):
                ejobs.append([HIPCC] + KERNEL_FLAGS + ["-c", src, "-o", obj])
        if ejobs:
            with cf.ThreadPoolExecutor(max_workers=min(6, len(ejobs))) as ex:
                for out in ex.map(_run, ejobs):
                    if verbose and out.strip():
                        print(out)
        elib = os.path.join(LIBDIR, "libmila_cdna4_experiments.so")
        if eobjs and (force or ejobs or _newer(elib, eobjs + [lib])):
            _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", elib] + eobjs + ["-L" + LIBDIR, "-lmila_cdna4", "-Wl,-rpath,$ORIGIN"])

    # C++ host mirror (template surface + model runners) -> libmila_host.so
    host_src = os.path.join(HOST, "src")
    if os.path.isdir(host_src):
        hh = _headers(HOST) + hdrs
        hobjs, hjobs = [], []
        for s in sorted(f for f in os.listdir(host_src) if f.endswith(".cpp")):
            src = os.path.join(host_src, s)
            obj = os.path.join(OBJDIR, "host_" + s[:-4] + ".o")
            hobjs.append(obj)
            if force or _newer(obj, [src] + hh):
                hjobs.append([HOSTCXX] + HOST_FLAGS + ["-c", src, "-o", obj])
        if hjobs:
            with cf.ThreadPoolExecutor(max_workers=min(6, len(hjobs))) as ex:
                for out in ex.map(_run, hjobs):
                    if verbose and out.strip():
                        print(out)
        hlib = os.path.join(LIBDIR, "libmila_host.so")
        if hobjs and (force or hjobs or _newer(hlib, hobjs + [lib])):
            _run([HOSTCXX, "-shared", "-fPIC", "-o", hlib] + hobjs +
                 ["-L" + LIBDIR, "-lmila_cdna4", "-L/opt/rocm/lib", "-lamdhip64", "-lrocprofiler-sdk-roctx", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"])
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    print("built", os.path.join(LIBDIR, "libmila_cdna4.so"))
