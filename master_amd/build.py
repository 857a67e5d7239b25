"""Builds libmi_pt.so (HIP kernels + C-ABI host code) for gfx950 with hipcc, in-tree.

The library is the product: there is no CPU fallback and no JIT cache.  Flags:
  -ffp-contract=off   fusion happens only where the sources spell fmaf() (arithmetic contract)
  -munsafe-fp-atomics LDS FP64 accumulation compiles to ds_add_f64, not a CAS loop
  -fno-slp-vectorize  no packed FP32 math (v_pk_fma_f32 issues at half rate on gfx950)
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmi_pt.so")

SOURCES = [
    "mi_pt_api.hip",
    "device/pt_kernels.hip",
    ("device/pt_kernels.hip", ["-DMI_PT_FAST=1"], "fastmath"),                                     # opt-in variant: v_rcp / v_rsq / v_sqrt / v_sin / v_cos instead of the IEEE forms
    "device/bvh_build.hip",
    "device/wf_kernels.hip",
    "device/bpt_kernels.hip",                                                                       # every BSDF, any beta
    ("device/bpt_kernels.hip", ["-DMI_BPT_FEAT=3", "-DMI_BPT_NS=bpt_fixed"], "bpt_fixed"),          # beta in {0, 1, 2}
    ("device/bpt_kernels.hip", ["-DMI_BPT_FEAT=0", "-DMI_BPT_NS=bpt_plain"], "bpt_plain"),          # ... and diffuse surfaces only
    "scene_host.cpp",
    "blend_reader.cpp",
    "exr_io.cpp",
]

HEADERS = [
    "../../include/mi_pt.h", "scene_host.hpp", "device/layout.h", "device/launch.h", "device/pt_device.h",
    "device/rng.h", "device/vecmath.h", "device/bpt.h", "device/wavefront.h",
]


def hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


BUILD_ID_TAG = b"MI_PT_BUILD_ID="


def source_hash(extra_flags=()):
    """sha256 over every source, header and flag that goes into the library: what `mi_pt_build_id()` of a library built from this tree returns."""
    import hashlib

    h = hashlib.sha256()
    files = sorted({os.path.normpath(os.path.join(CSRC, s if isinstance(s, str) else s[0])) for s in SOURCES + HEADERS} | {os.path.abspath(__file__)})
    for f in files:
        h.update(os.path.relpath(f, HERE).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    h.update(repr([e if isinstance(e, str) else list(e[:2]) + [e[2]] for e in SOURCES]).encode())
    h.update(repr(list(extra_flags)).encode())
    h.update(toolchain_id().encode())  # a library built by another compiler / ROCm release is stale too
    return h.hexdigest()[:32]


_TOOLCHAIN = None


def toolchain_id():
    """`hipcc --version` of the compiler build() would use (the same image on the GPU box gives the same text), or its path when it cannot be run."""
    global _TOOLCHAIN
    if _TOOLCHAIN is None:
        try:
            cc = hipcc()
            _TOOLCHAIN = subprocess.run([cc, "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode(errors="replace")
        except Exception as e:  # noqa: BLE001
            _TOOLCHAIN = "hipcc unavailable: %r" % (e,)
    return _TOOLCHAIN


def library_build_id(path=None):
    """The id embedded in a built library (read from the bytes of the file, the library is not loaded), or None."""
    path = path or LIB
    try:
        import mmap

        with open(path, "rb") as fh, mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ) as blob:  # scanned in place, not read into memory
            i = blob.find(BUILD_ID_TAG)
            if i < 0:
                return None
            j = i + len(BUILD_ID_TAG)
            return blob[j:j + 32].decode("ascii", "replace")
    except (OSError, ValueError):
        return None


def needs_build(extra_flags=()):
    """True unless the library on disk was built from exactly these sources and flags.  Timestamps are not trusted: a prebuilt *.so travels to the
    GPU box next to sources that may have changed since."""
    return library_build_id() != source_hash(extra_flags)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """out: alternative output path (A/B variants, e.g. build(extra_flags=["-DX"], out="libmi_pt_x.so"))."""
    global LIB
    if out is None and not force and not needs_build(extra_flags):
        return LIB
    lib_saved = LIB
    if out is not None:
        LIB = os.path.join(HERE, out)
    try:
        return _build(verbose, extra_flags)
    finally:
        LIB = lib_saved


def _build(verbose, extra_flags):
    objdir = os.path.join(HERE, "build" + ("_" + os.path.basename(LIB) if not LIB.endswith("libmi_pt.so") else ""))
    os.makedirs(objdir, exist_ok=True)
    common = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
              "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include")]
    objs = []
    procs = []
    for entry in SOURCES:
        src, variant_flags, tag = entry if isinstance(entry, tuple) else (entry, [], "")
        obj = os.path.join(objdir, src.replace("/", "_") + ("." + tag if tag else "") + ".o")
        cmd = [hipcc()] + common + list(variant_flags)
        if src.endswith(".hip"):
            # -fno-slp-vectorize: the SLP vectoriser packs pairs of dot products into v_pk_fma_f32 / v_pk_mul_f32, which issue at half the rate of
            # the scalar form on gfx950 and need v_mov / v_pk_mov to line their operands up: C2 +15 % without it (profiles/r03/ab_flat.txt)
            cmd += ["--offload-arch=gfx950", "-munsafe-fp-atomics", "-fno-slp-vectorize"]
        else:
            cmd += ["-x", "c++"]
        if src == "scene_host.cpp":  # mi_pt_build_id(): the hash of what this library is built from
            cmd += ['-DMI_PT_BUILD_ID_STRING="%s%s"' % (BUILD_ID_TAG.decode(), source_hash(extra_flags))]
        cmd += list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"))
    cmd = [hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-lz"]  # zlib: ZIP-compressed EXR input (exr_io.cpp)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
