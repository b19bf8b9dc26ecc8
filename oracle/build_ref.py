#!/usr/bin/env python
"""oracle/build_ref.py -- TEST INFRASTRUCTURE.  Builds oracle/_ref/libadcensus_ref.so:
the reference's own adcensus.cu, unmodified and read from /root/reference where it lies,
compiled for gfx950 by hipcc through the fake Lua/TH/THC headers of oracle/ref_stubs/
(see oracle/ref_shim.hip).  Only runs where /root/reference exists; the .so is git-ignored
and travels to the GPU box with the repo snapshot.

Flags: -O3 -DNDEBUG as the reference's Makefile.proto:9 (`nvcc -arch sm_35 -O3 -DNDEBUG`);
-ffp-contract=fast so that `a*b+c` contracts to an FMA the way nvcc's default --fmad=true
does; -fno-slp-vectorize because hipcc's SLP pass otherwise packs `sum += a*b; cnt += b`
(mean2d, adcensus.cu:1254-1255) into v_pk_add_f32 and thereby blocks that contraction,
which nvcc (no packed fp32) performs.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("MC_REFERENCE_DIR", "/root/reference")
OUT = os.path.join(HERE, "_ref", "libadcensus_ref.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False):
    src = os.path.join(REF, "adcensus.cu")
    if not os.path.exists(src):
        print("build_ref: %s not found; keeping any prebuilt %s" % (src, OUT))
        return os.path.exists(OUT)
    deps = [src, os.path.join(REF, "SpatialLogSoftMax.cu"), os.path.join(HERE, "ref_shim.hip")]
    deps += [os.path.join(HERE, "ref_stubs", f) for f in os.listdir(os.path.join(HERE, "ref_stubs")) if f.endswith(".h")]
    deps.append(os.path.join(HERE, "ref_stubs", "png++", "image.hpp"))
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return True
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-DNDEBUG", "-ffp-contract=fast", "-fno-slp-vectorize", "-fPIC",
           "-shared", "-fvisibility=hidden", "-Wno-format-security", "-x", "hip",
           "-I", os.path.join(HERE, "ref_stubs"), "-I", REF, os.path.join(HERE, "ref_shim.hip"), "-o", OUT]
    subprocess.check_call(cmd)
    return True


if __name__ == "__main__":
    sys.exit(0 if build(force="--force" in sys.argv) else 1)
