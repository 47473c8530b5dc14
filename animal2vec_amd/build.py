"""Builds liba2v_hip.so (hipcc, gfx950 only) in-tree.  No JIT cache: the .so travels with the repo."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "liba2v_hip.so")
SOURCES = ["a2v_gemm.hip", "a2v_gemm_fast.hip", "a2v_gemm_p8.hip", "a2v_gemm_tn8.hip", "a2v_wgrad_slab.hip", "a2v_wgrad_sparse.hip", "a2v_conv_slab.hip", "a2v_conv_sparse.hip", "a2v_conv_dec.hip", "a2v_attn.hip", "a2v_norm.hip", "a2v_elem.hip", "a2v_sinc.hip", "a2v_mixup.hip", "a2v_mask.hip", "a2v_conv_strided.hip", "a2v_split.hip"]
# -amdgpu-mfma-vgpr-form: MFMA accumulators stay in arch VGPRs (no kernel here needs > 256 registers), which removes the
# v_accvgpr_read/write copies around every VALU use of an accumulator (softmax, epilogues: 96 of 417 VALU per attention step)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("A2V_HIPCC_FLAGS", "").split()      # extra -D switches of experiment builds (same-box A/B with tools/build_prev.sh)
# a2v_attn.hip without the SLP vectoriser: paired into v_pk_fma_f32 the ALiBi term loses the |.| source modifier (one extra
# v_and per score), and packed fp32 ops cost more issue time between MFMAs than two plain ones
FILE_FLAGS = {"a2v_attn.hip": ["-fno-slp-vectorize"]}
# sources that only experiment libraries carry, by the -D switch that asks for them (never part of the product build):
#   python -m animal2vec_amd.build --variant w2 -DA2V_W2     -> liba2v_hip_w2.so with the 256 x 128 two-per-CU NT kernel (round 4's
#   negative result, profiles/r04_w2_experiment.txt), switched on with a2v_debug_set(25, 2)
VARIANT_SOURCES = {"-DA2V_W2": ["a2v_gemm_w2.hip"]}


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "a2v_common.h"), os.path.join(CSRC, "a2v_gemm.h"), os.path.join(CSRC, "a2v_attn_seq2.h")]
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(OBJ, s.replace(".hip", ".o"))
        if force or _stale(obj, [src, *hdrs]):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    build_abi_client(force=force or bool(jobs), verbose=verbose)
    # the fp16 twin (csrc/a2v_common.h: A2V_ELT_F16): same sources, IEEE half as the 16-bit type -- the EMA teacher's forward
    build_variant("f16", ["-DA2V_ELT_F16"], verbose=verbose)
    return LIB


def build_abi_client(force: bool = False, verbose: bool = True) -> str:
    """tests/abi_client: a torch-free C++ program against include/a2v.h + liba2v_hip.so (run by the GPU tests)."""
    root = os.path.dirname(HERE)
    src = os.path.join(root, "tests", "abi_client", "abi_client.cpp")
    out = os.path.join(root, "tests", "abi_client", "abi_client.bin")
    if not os.path.exists(src):
        return ""
    if force or _stale(out, [src, os.path.join(root, "include", "a2v.h"), LIB]):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(root, "include"), src, "-o", out,
               "-L", HERE, "-la2v_hip", "-Wl,-rpath,$ORIGIN/../../animal2vec_amd"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f"abi client build failed:\n{r.stderr}")
    return out


def build_variant(tag: str, defines, verbose: bool = False) -> str:
    """Experiment / diagnostic builds next to the product library (liba2v_hip_<tag>.so, own object directory): same sources with
    extra -D switches, e.g. build_variant("stamp", ["-DP8_STAMP"]) for tools/inkernel_clock.py.  Selected at run time with
    A2V_LIB_PATH; never loaded by default, never built by build()."""
    obj_dir = os.path.join(CSRC, "build_" + tag)
    os.makedirs(obj_dir, exist_ok=True)
    lib = os.path.join(HERE, f"liba2v_hip_{tag}.so")
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, "a2v_common.h"), os.path.join(CSRC, "a2v_gemm.h"), os.path.join(CSRC, "a2v_attn_seq2.h")]

    def cc(s):
        src, obj = os.path.join(CSRC, s), os.path.join(obj_dir, s.replace(".hip", ".o"))
        if _stale(obj, [src, *hdrs]):
            cmd = [hipcc, *FLAGS, *defines, *FILE_FLAGS.get(s, []), "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        return obj

    sources = SOURCES + [f for d in defines for f in VARIANT_SOURCES.get(d, [])]
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, sources))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:          # python -m animal2vec_amd.build --variant stamp -DP8_STAMP
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build(force="--force" in sys.argv))
