"""CPU: the C-ABI library loads and exports every symbol include/a2v.h declares (no compute)."""
import os
import re

import pytest

from conftest import ROOT


def header_decls():
    txt = open(os.path.join(ROOT, "include", "a2v.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(?:int|size_t)\s+(a2v_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = n
    return decls


def test_library_exports_every_declared_symbol():
    from animal2vec_amd import _lib
    lib = _lib.load()
    decls = header_decls()
    assert len(decls) >= 25
    for name, nargs in decls.items():
        assert hasattr(lib, name), f"{name} declared in include/a2v.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
        assert len(_lib.SIGNATURES[name]) == nargs, (name, len(_lib.SIGNATURES[name]), nargs)
    assert set(_lib.SIGNATURES) == set(decls)
    assert lib.a2v_abi_version() == 14


def test_fp16_twin_library_exports_the_same_abi():
    """liba2v_hip_f16.so (csrc/a2v_common.h: A2V_ELT_F16; the EMA teacher's opt-in fp16 forward): the same sources, the same C ABI."""
    from animal2vec_amd import _lib
    lib = _lib.load_f16()
    for name in header_decls():
        assert hasattr(lib, name), f"{name} missing from liba2v_hip_f16.so"
    assert lib.a2v_abi_version() == _lib.load().a2v_abi_version()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from animal2vec_amd import ops, _lib
    with pytest.raises(_lib.A2VError):
        ops.linear(torch.zeros(8, 8), torch.zeros(8, 8))


def _oracle_uses(path):
    """Syntax-tree walk (a grep is defeated by "a2v_" + "oracle"): every import statement, every dynamic import call
    (importlib.import_module / __import__ must take ONE literal argument) and every sys.path / os.path.join literal."""
    import ast
    bad = []
    banned = ("a2v_oracle", "ref_shims", "gen_golden", "oracle")
    tree = ast.parse(open(path).read(), filename=path)

    def is_banned(name):
        return any(part in banned for part in name.split("."))

    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            bad += [f"import {a.name}" for a in node.names if is_banned(a.name)]
        elif isinstance(node, ast.ImportFrom):
            if node.module and is_banned(node.module):
                bad.append(f"from {node.module} import ...")
            bad += [f"from {node.module} import {a.name}" for a in node.names if is_banned(a.name)]
        elif isinstance(node, ast.Call):
            f = node.func
            fname = f.attr if isinstance(f, ast.Attribute) else (f.id if isinstance(f, ast.Name) else "")
            if fname in ("import_module", "__import__"):
                a0 = node.args[0] if node.args else None
                if not (isinstance(a0, ast.Constant) and isinstance(a0.value, str)):
                    bad.append(f"dynamic import with a computed module name (line {node.lineno})")
                elif is_banned(a0.value):
                    bad.append(f"dynamic import of {a0.value}")
            if fname in ("exec", "eval", "run_path", "run_module", "spec_from_file_location"):
                bad.append(f"{fname}() call (line {node.lineno})")
        elif isinstance(node, ast.Constant) and isinstance(node.value, str):
            v = node.value
            if "\n" not in v and len(v) < 64 and (v in banned or v.startswith("oracle/") or v.endswith("/oracle")):
                bad.append(f"string literal {v!r} (line {node.lineno})")
    return bad


def test_product_never_imports_oracle():
    """The product (the package, the `nn` alias, the training entry point) never reaches the oracle.  bench.py is checked
    too, except its cpu_baseline leg (`_cpu_leg`), the one place the brief allows it."""
    import ast
    files = []
    for top in ("animal2vec_amd", "nn"):
        for dp, _, fs in os.walk(os.path.join(ROOT, top)):
            files += [os.path.join(dp, f) for f in fs if f.endswith(".py")]
    files.append(os.path.join(ROOT, "animal2vec_train.py"))
    for path in files:
        assert not _oracle_uses(path), (path, _oracle_uses(path))
    assert not os.path.exists(os.path.join(ROOT, "animal2vec_amd", "smoke.py"))
    # bench.py: only _cpu_leg may touch oracle/
    bpath = os.path.join(ROOT, "bench.py")
    tree = ast.parse(open(bpath).read())
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name == "_cpu_leg":
            continue
        seg = ast.Module(body=[node], type_ignores=[])
        tmp = os.path.join(ROOT, "tests", "__pycache__", "_bench_seg.py")
        os.makedirs(os.path.dirname(tmp), exist_ok=True)
        with open(tmp, "w") as f:
            f.write(ast.unparse(seg))
        assert not _oracle_uses(tmp), (node.lineno, _oracle_uses(tmp))


def test_oracle_guard_catches_split_strings(tmp_path):
    p = tmp_path / "x.py"
    p.write_text('import importlib\nO = importlib.import_module("a2v_" + "oracle")\n')
    assert _oracle_uses(str(p))
    p.write_text('import sys\nsys.path.insert(0, "oracle")\nimport a2v_oracle\n')
    assert len(_oracle_uses(str(p))) == 2
    p.write_text('import numpy as np\nx = "the oracle says"\n')
    assert not _oracle_uses(str(p))


def test_p8_tile_claim_register_is_untouched_until_its_wait(tmp_path):
    """The persistent 8-phase NT GEMM claims the tile after the next one with a raw-asm returning atomic at the top of the epilogue;
    the result lands in a VGPR asynchronously and is retired by a hand-counted `s_waitcnt vmcnt(N)` (N = 8 ... 14 by variant) in front of the slot
    publication at the end of the epilogue.  hipcc does not know the register is in flight: if it splits that value's live range,
    a v_mov reads the register BEFORE the atomic has returned and moves the stale copy back afterwards -- tile ids repeat and the
    launch never ends (seen once, with the claim still issued two K-tiles before the seam, when the 256-row tile's epilogue grew by
    eight live registers).  So the ISA of every variant is checked: the first instruction that names the claim's destination
    register after the atomic must come after that counted wait."""
    import shutil
    import subprocess
    import pytest
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    checked = 0
    src = "a2v_gemm_p8.hip"
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    for m in re.finditer(r"^(_Z\d+gemm_nt_bf16_p8_kernel\w*):", s, re.M):
        body = s[m.end():s.index("s_endpgm", m.end())]
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        mfma = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        for n, t in enumerate(lines):
            mm = re.match(r"global_atomic_add (v\d+), v\d+, v\d+, s\[\d+:\d+\] sc0$", t)
            if not mm or not mfma or n < mfma[0]:
                continue                                   # (the claims in front of the tile loop are waited for at once)
            reg = mm.group(1)
            first_use = next(i for i, l in enumerate(lines) if i > n and re.search(r"\b" + reg + r"\b", l))
            waits = [i for i in range(n + 1, first_use) if re.fullmatch(r"s_waitcnt vmcnt\((8|10|12|14)\)", lines[i])]
            assert waits, (m.group(1), reg, n, first_use, lines[first_use])
            checked += 1
        # The K loop keeps three half-tiles of LDS-DMA in flight behind hand-counted waits: a compiler-placed s_waitcnt vmcnt(0)
        # or a spill reload (a scratch load, which draws one) between the tile's first and last MFMA drains that pipeline at
        # every tile -- the cost of several of the regressions fixed in round 2 (conditional loads, spilled lane constants).
        drains = [(i, lines[i]) for i in range(mfma[0], mfma[-1])
                  if lines[i].startswith("s_waitcnt vmcnt(0)") or lines[i].startswith("scratch_")]
        assert not drains, (m.group(1), drains[:4])
        # the direct-store epilogue waits for its loads with counted waits only: vmcnt(0) stays at its top (bias) and at the
        # kernel's end -- a handful, not one per row block
        o32 = "ELb1EEv" in m.group(1)        # fp32-output (bf16x3) instantiations: plain C++ epilogue, compiler-placed waits
        if "ELb1ELi" in m.group(1) and not o32:
            tail0 = [i for i in range(mfma[-1], len(lines)) if lines[i].startswith("s_waitcnt vmcnt(0)")]
            assert len(tail0) <= 5, (m.group(1), len(tail0))
    assert checked == 31         # three tile heights x (two epilogue forms + the fp32-output form) + eleven fixed flavours x two tile heights
    # the fixed-flavour instantiations exist because the generic 256-row epilogue has no registers to spare: none of them may spill
    for m in re.finditer(r"\.amdhsa_kernel (_Z\d+gemm_nt_bf16_p8_kernelILi[34]ELb1ELi([1-9]\d*)E\w*)(.*?)\.end_amdhsa_kernel", s, re.S):
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", m.group(3)).group(1))
        assert scratch == 0, (m.group(1), scratch)


def _min_vmem_to_back_edge(lines, start):
    """The smallest number of vector-memory instructions that are certain to issue (not under a lane mask) on any control-flow path from
    lines[start] to a backward branch that re-enters the code in front of `start` (the tile loop's back edge).  Paths that leave the
    loop do not count.  Shortest path over the instruction graph (fall-through + branch targets)."""
    import heapq
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r"(\.LBB\w+):", l)] if m}
    # lane-mask state is tracked along the text (the epilogue's masked regions are properly nested if / endif pairs)
    masked, m = [False] * len(lines), False
    for i, l in enumerate(lines):
        if re.match(r"s_(and|andn2|xor|or)_saveexec_b64|v_cmpx_", l) or re.match(r"s_(and|andn2)_b64 exec,", l):
            m = True
        elif re.match(r"s_or_b64 exec, exec|s_mov_b64 exec,", l):
            m = False
        masked[i] = m
    cost = lambda i: 1 if (re.match(r"(global|buffer|flat)_(load|store|atomic)", lines[i]) and not masked[i]) else 0
    best, dist, heap = None, {start: 0}, [(0, start)]
    while heap:
        d, i = heapq.heappop(heap)
        if d > dist.get(i, 1 << 30) or i >= len(lines):
            continue
        d2 = d + cost(i)
        mm = re.match(r"s_cbranch_\w+ (\.LBB\w+)|s_branch (\.LBB\w+)", lines[i])
        succ = []
        if mm:
            t = labels[mm.group(1) or mm.group(2)]
            if t < start:
                best = d2 if best is None else min(best, d2)        # the back edge: this path is complete
            else:
                succ.append(t)
            if mm.group(1):
                succ.append(i + 1)
        else:
            succ.append(i + 1)
        for j in succ:
            if j < len(lines) and d2 < dist.get(j, 1 << 30):
                dist[j] = d2
                heapq.heappush(heap, (d2, j))
    return best


def test_p8_fixed_flavour_seam_wait_is_covered_by_the_epilogue(tmp_path):
    """ADVICE r4 (medium).  In the fixed-flavour instantiations the first K-tile of a tile waits with vmcnt(6 + E) (P8_KTILE_FIRST): the
    six LDS-DMA operations of its own phases 2-4 plus E operations of the epilogue in front of it may stay in flight, everything older
    -- K-tile 1 of the tile, staged before the epilogue -- must be complete.  E is derived by hand from the epilogue's source (C / C2
    stores, bias loads, residual loads); if hipcc ever emitted FEWER vector-memory operations than that on some path (merged or
    predicated stores, another lowering of the 16-B vectors), the wait would let K-tile 1's DMA be in flight when its buffer is read:
    wrong GEMM output, silently.  So the ISA is counted: between the seam's last counted wait (vmcnt(8), P8_KTILE_LAST) and the end
    of the tile loop's body, at least min(E, 48) vector-memory instructions must be issued unconditionally, and the first K-tile's wait
    must be exactly vmcnt(6 + min(E, 48))."""
    import shutil
    import subprocess
    import pytest
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = "a2v_gemm_p8.hip"
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    F_BIAS, F_RES, F_GELU_BWD, F_STORE_PRE = 1, 4, 8, 16
    seen = 0
    for m in re.finditer(r"^(_Z\d+gemm_nt_bf16_p8_kernelILi([34])ELb1ELi([1-9]\d*)ELb0EE\w*):", s, re.M):
        ih, fx = int(m.group(2)), int(m.group(3)) & 0xFF
        e = 2 * (4 + ih) * (2 if fx & F_STORE_PRE else 1) + (4 if fx & F_BIAS else 0) + (2 * (4 + ih) if fx & (F_RES | F_GELU_BWD) else 0)
        e = min(e, 48)
        body = s[m.end():s.index("s_endpgm", m.end())]
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        w8 = [i for i, l in enumerate(lines) if l == "s_waitcnt vmcnt(8)"]
        w0 = [i for i, l in enumerate(lines) if l == f"s_waitcnt vmcnt({6 + e})"]
        assert len(w8) == 1, (m.group(1), w8)
        assert w0, (m.group(1), 6 + e, sorted({l for l in lines if l.startswith("s_waitcnt vmcnt")}))
        n = _min_vmem_to_back_edge(lines, w8[0])
        assert n is not None and n >= e, (m.group(1), "epilogue issues", n, "certain vector-memory operations, the seam wait assumes", e)
        seen += 1
    assert seen == 22            # eleven fixed flavours x two tile heights


def test_conv_dec_counted_fetch_registers_are_untouched_until_their_wait(tmp_path):
    """a2v_conv_dec.hip (round 5): the next slab is fetched by raw-asm loads into AGPRs and waited for by a hand-counted
    `s_waitcnt vmcnt(N)` in front of the stash, so that the wait does not sit out the acknowledgement of the block's output stores.
    hipcc treats an asm output as available at once: with VGPR destinations it copied every loaded register into an AGPR right behind
    the load (v_accvgpr_write of data that had not arrived).  With AGPR destinations there is no copy -- checked here for every
    instantiation: no instruction between the first load of the block loop and the counted wait names a destination register."""
    import shutil
    import subprocess
    import pytest
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = "a2v_conv_dec.hip"
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    seen = 0
    for m in re.finditer(r"^(_Z15conv_dec_kernelILi\d+ELi\d+ELi\d+EE\w*):", s, re.M):
        body = s[m.end():s.index("s_endpgm", m.end())]
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        fb = [i for i, l in enumerate(lines) if l == "s_barrier"][0]
        loads = [(i, re.match(r"global_load_dwordx4 a\[(\d+):(\d+)\]", l)) for i, l in enumerate(lines) if i > fb]
        loads = [(i, int(mm.group(1)), int(mm.group(2))) for i, mm in loads if mm]
        assert len(loads) in (7, 9, 14), (m.group(1), len(loads))
        regs = set()
        for _, a, b in loads:
            regs |= set(range(a, b + 1))
        waits = [i for i, l in enumerate(lines) if i > loads[-1][0] and re.fullmatch(r"s_waitcnt vmcnt\((8|12|16|24|32)\)", l)]
        assert waits, m.group(1)
        for i in range(loads[0][0], waits[0]):
            l = lines[i]
            if l.startswith("global_load_dwordx4 a["):
                continue
            named = set()
            for mm in re.finditer(r"\ba(\d+)\b|a\[(\d+):(\d+)\]", l):
                named |= {int(mm.group(1))} if mm.group(1) else set(range(int(mm.group(2)), int(mm.group(3)) + 1))
            assert not (named & regs), (m.group(1), i, l)
        seen += 1
    assert seen == 3


def test_attn_fwd32_tile_dma_is_not_waited_for_before_the_tile_math(tmp_path):
    """a2v_attn.hip, attn_fwd32_bf16_kernel (round 5): the next K / V tile is fetched by LDS-DMA at the top of a key tile and must
    land UNDER that tile's math; the only wait for it belongs in front of the barrier at the tile's end.  hipcc inserts
    `s_waitcnt vmcnt(0)` in front of LDS reads it believes may alias a pending LDS-DMA -- it did for static __shared__ arrays,
    for the ds_read_tr builtin and for a 64-byte vector load, which parked every wave for the global latency once per tile.  The
    kernel avoids those forms (dynamic LDS, raw-asm transposing reads, 16-byte loads); this test holds the result: in every key-tile
    loop of every instantiation, no vmcnt wait between the tile's DMA instructions and its LAST MFMA, the raw reads' counted waits
    present (lgkmcnt(3) for the K ring, lgkmcnt(4) for the V steps), and no spill."""
    import shutil
    import subprocess
    import pytest
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = "a2v_attn.hip"
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    seen = loops = 0
    for m in re.finditer(r"^(_Z22attn_fwd32_bf16_kernelILb[01]ELb[01]ELb[01]EE\w*):", s, re.M):
        name = m.group(1)
        body = s[m.end():s.index("s_endpgm", m.end())]
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
        meta = s[s.index(".name:           " + name):]
        assert int(re.search(r"\.vgpr_spill_count:\s*(\d+)", meta).group(1)) == 0, name
        dma = [i for i, l in enumerate(lines) if l.startswith("global_load_lds_dwordx4")]
        bars = [i for i, l in enumerate(lines) if l == "s_barrier"]
        # groups of four DMA instructions; the prologue's group is followed by a barrier before any MFMA
        groups = [dma[i:i + 4] for i in range(0, len(dma), 4)]
        assert all(len(g) == 4 and g[3] - g[0] < 40 for g in groups), (name, dma)
        for g in groups:
            nb = min(b for b in bars if b > g[3])
            seg = lines[g[3]:nb]
            mf = [i for i, l in enumerate(seg) if l.startswith("v_mfma_f32_32x32x16_bf16")]
            if not mf:
                continue                                   # the prologue's fetch
            assert len(mf) in (16, 20), (name, len(mf))
            for l in seg[:mf[-1]]:
                assert "vmcnt" not in l, (name, l)
            assert sum(1 for l in seg if l == "s_waitcnt lgkmcnt(3)") >= 5, name
            assert sum(1 for l in seg if l == "s_waitcnt lgkmcnt(4)") == 3, name
            assert sum(1 for l in seg if l.startswith("ds_read_b64_tr_b16")) == 16, name
            loops += 1
        seen += 1
    assert seen == 5 and loops >= 15, (seen, loops)


def test_attn_seq2_fetches_are_not_waited_for_inside_the_math(tmp_path):
    """a2v_attn_seq2.h (round 6): the persistent student-attention kernels fetch the NEXT sequence's images by LDS-DMA at the top of a
    sequence (forward) / of a role (backward); they must land under the current math.  The traps, each of which once serialised the
    pipeline: a use of a fetched register value (hipcc waits vmcnt(0)), the ds_read_tr builtin (vmcnt(0) in front of it: "may alias
    the DMA"), an LDS atomicAdd (vmcnt(0) in front of every ds_add), a spilled register (its reload waits vmcnt(0)).  Held here:
    no scratch, no SGPR-spill-free requirement on the forward, and in every kernel the ONLY vmcnt waits inside the sequence loop are
    the hand-placed `s_waitcnt vmcnt(0)` (asm) -- at most three per loop body -- with every transposed read inside an asm block."""
    import shutil
    import subprocess
    import pytest
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = "a2v_attn.hip"
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    seen = 0
    for m in re.finditer(r"^(_Z20attn_seq2_(fwd|bwd)_kernelI\w*):", s, re.M):
        name, kind = m.group(1), m.group(2)
        body = s[m.end():s.index("s_endpgm", m.end())]
        meta = s[s.index(".name:           " + name):]
        assert int(re.search(r"\.vgpr_spill_count:\s*(\d+)", meta).group(1)) == 0, name
        assert "scratch_" not in body, name
        lines = [l.strip() for l in body.split("\n") if l.strip()]
        # the loop over sequences: from its header to the end of the kernel
        hdr = [i for i, l in enumerate(lines) if "Loop Header: Depth=1" in l]
        assert hdr, name
        loop = lines[hdr[0]:]
        in_asm = False
        compiler_vm, asm_vm, tr_outside = [], 0, 0
        for l in loop:
            if l.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if l.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if l.startswith(";"):
                continue
            if "vmcnt" in l:
                if in_asm:
                    asm_vm += 1
                else:
                    compiler_vm.append(l)
            if l.startswith("ds_read_b64_tr_b16") and not in_asm:
                tr_outside += 1
            assert not l.startswith(("ds_add_f32", "ds_add_rtn_f32")), (name, l)
        assert tr_outside == 0, (name, tr_outside)
        assert 1 <= asm_vm <= 4, (name, asm_vm)
        # what the compiler adds on its own may only sit directly behind one of those hand-placed waits (the register copies of the
        # fetched values at the top of the next sequence): never a wait the math has to sit behind
        assert len(compiler_vm) <= 4, (name, compiler_vm)
        seen += 1
    assert seen >= 8, seen


def test_tn8_kernel_has_no_spills_and_no_drain_at_unit_seams(tmp_path):
    """256 x 256 TN weight-gradient kernel: at the register limit (256) the lane constants of the staging addresses used to be
    spilled and reloaded at the start of every work unit behind `s_waitcnt vmcnt(0)` -- a drain of the previous unit's 128
    atomics per wave.  They come from the hardware lane id now: no scratch, and the only vmcnt(0) is the one before the kernel ends."""
    import shutil
    import subprocess
    import pytest
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = "a2v_gemm_tn8.hip"
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    m = re.search(r"^(_Z\d+gemm_tn_bf16_p8_kernel\w*):", s, re.M)
    assert m
    body = s[m.end():s.index("s_endpgm", m.end())]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith(";")]
    assert not [l for l in lines if l.startswith("scratch_")]
    drains = [i for i, l in enumerate(lines) if l.startswith("s_waitcnt vmcnt(0)")]
    assert len(drains) == 1 and drains[0] >= len(lines) - 3, drains


HOT_KERNELS = {
    "a2v_attn.hip": [r"attn_fwd_bf16_kernelILi64ELb0ELi2ELb[01]ELb0E", r"attn_fwd32_bf16_kernel", r"attn_fwd_seq48_kernel", r"attn_bwd_seq48_kernel"],
    "a2v_mask.hip": [r"span_mask_rows_kernel", r"span_mask_finish_kernel"],
    "a2v_conv_slab.hip": [r"conv_nt_slab2_kernel"],
    "a2v_wgrad_slab.hip": [r"conv_wgrad_slab_kernel"],
    "a2v_conv_dec.hip": [r"conv_dec_kernel"],
    "a2v_norm.hip": [r"layernorm_fwd_kernelIDF16bLi2E", r"layernorm_bwd_kernelIDF16bLi2ELi8ELi[01]ELi0E",
                     r"layernorm_bwd_kernelIDF16bLi2ELi8ELi0ELi2E", r"instnorm_"],
    "a2v_elem.hip": [r"d2v_loss_kernel", r"decoder_input_kernel", r"col2im_kernel", r"adam_kernel", r"ema_kernel"],
    # (the four-wave form of sinc_fused_bwd needed 512 registers and spilled 7: 0.06 of its roofline)
    "a2v_sinc.hip": [r"sinc_fused_fwd_kernel", r"sinc_fused_bwd_kernel"],
    # round 5: the row-sparse pos-conv products (232-246 registers each: accumulators / weight fragments fill the budget)
    "a2v_wgrad_sparse.hip": [r"conv_wgrad_sparse_kernelILb1E", r"conv_wgrad_sparse_kernelILb0E"],
    "a2v_conv_sparse.hip": [r"conv_sparse_in_kernelILb1E", r"conv_sparse_in_kernelILb0E"],
}


@pytest.mark.parametrize("src", sorted(HOT_KERNELS))
def test_hot_kernels_do_not_spill(src, tmp_path):
    """The kernels that make up the pretraining step's device time are written against fixed register budgets (launch bounds,
    accumulators in registers); a spill in one of them is a silent slowdown -- scratch traffic, and a `s_waitcnt vmcnt(0)` in front
    of every reload that drains whatever prefetch is in flight.  Compiled with the library's own flags, each must use no scratch."""
    import shutil
    import subprocess
    from animal2vec_amd import build as B
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path / (src + ".s")
    cmd = [hipcc, *B.FLAGS, *B.FILE_FLAGS.get(src, []), "-S", "--cuda-device-only", os.path.join(B.CSRC, src), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    s = out.read_text()
    seen = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        if not any(re.search(pat, m.group(1)) for pat in HOT_KERNELS[src]):
            continue
        seen += 1
        scratch = int(re.search(r"\.amdhsa_private_segment_fixed_size\s+(\d+)", m.group(2)).group(1))
        assert scratch == 0, (m.group(1), scratch)
    assert seen >= len(HOT_KERNELS[src])
