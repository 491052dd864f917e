"""Build-time check of the hand-counted LDS-DMA kernels: compiles flash_prefill.hip to gfx950 ISA and asserts, for every
instantiation, what the hand-written s_waitcnt vmcnt counts rely on for SPEED (they stay correct with extra vector-memory operations:
an extra operation can only make a counted wait stricter, never looser — but a spill or a hoisted load inside the step loop silently
costs the ring its depth):
  * no scratch (spill) traffic anywhere in the kernel, no spilled registers in its metadata;
  * inside the basic blocks that issue MFMAs (the step loop) the only vector-memory instructions are the LDS-DMA requests and, in the
    paged forms, nothing else (the block-table reload is its own block);
  * every s_waitcnt vmcnt in those blocks is one of the hand-written ones (immediate 0 or a multiple of the pieces per tile).
    python tools/check_kernel_isa.py          exit code 0 = as expected; prints one line per instantiation"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_isa(src, flags=()):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = [HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", *flags, "-x", "hip", "-S", "--cuda-device-only", src, "-o", out]
    subprocess.run(cmd, check=True, cwd=os.path.join(ROOT, "nano-vllm-rs_amd", "csrc"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text, prefix):
    """{mangled name: list of basic blocks (lists of instruction lines)}"""
    res = {}
    for m in re.finditer(r"^(_Z\w*%s\w*):[^\n]*\n(.*?)s_endpgm" % prefix, text, re.S | re.M):
        blocks, cur = [], []
        for line in m.group(2).split("\n"):
            line = line.strip()
            if re.match(r"^\.LBB\d+_\d+:", line):
                blocks.append(cur); cur = []
            elif line and not line.startswith((";", ".")):
                cur.append(line)
        blocks.append(cur)
        res[m.group(1)] = blocks
    return res


def check_flash(verbose=True):
    text = compile_isa("kernels/flash_prefill.hip")
    bad = []
    meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
        r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)}
    ks = kernels(text, "flash_prefill_kernel")
    if len(ks) < 12:
        bad.append(f"only {len(ks)} flash_prefill_kernel instantiations found")
    for name, blocks in ks.items():
        problems = []
        flat = [i for b in blocks for i in b]
        if any(i.startswith("scratch_") for i in flat):
            problems.append("scratch traffic")
        if name in meta and (meta[name][0] or meta[name][1]):
            problems.append(f"spills {meta[name]}")
        dma = 0
        for b in blocks:
            if not any(i.startswith("v_mfma") for i in b):
                continue
            for i in b:
                if i.startswith(("global_load", "buffer_load", "global_store", "buffer_store", "flat_")) and not i.startswith("global_load_lds"):
                    problems.append("vector memory inside an MFMA block: " + i.split()[0])
                m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", i)
                if m and int(m.group(1)) not in (0, 4, 8, 16, 32):
                    problems.append("unexpected counted wait " + i)
        dma = sum(1 for i in flat if i.startswith("global_load_lds"))
        if dma == 0:
            problems.append("no LDS-DMA requests")
        if verbose:
            print(f"{name[:90]:90s} {'OK' if not problems else '; '.join(sorted(set(problems)))}  ({dma} LDS-DMA requests)")
        bad += [name + ": " + p for p in sorted(set(problems))]
    return bad


def check_attention_merge_forms(verbose=True):
    """r05: the cross-wave merge of attn_rows_kernel and merge_partitions are inlined into kernels whose results are promised to agree bit for bit
    (partial-writing form + attn_merge_kernel == last-arriver form == own-partition form).  Under -ffp-contract=fast hipcc chooses per INSTANTIATION
    whether "a += w * b" becomes an fma; the source now spells both out (mul_then_add with contraction off; explicit fmas in merge_partitions).  This
    check reads the ISA: no fused multiply-add in the cross-wave merge (between the first two workgroup barriers) of every instantiation that does not
    divide there, and no separate add anywhere in the merge kernel (the merges inlined into attn_rows_kernel share its source and are compared with it at
    scale by tests/test_engine_gpu.py and tests/test_kernels_gpu.py: block layout makes their region unreliable to cut out of the ISA)."""
    bad = []
    for flags in ((),):                                                     # (the bfloat16 build shares every line of it; one compile is ~100 s)
        text = compile_isa("kernels/attention.hip", flags)
        ks = kernels(text, "attn_rows_kernel")
        if len(ks) < 20:  # (fewer than the launch table can ask for)
            bad.append(f"only {len(ks)} attn_rows_kernel instantiations found")
        for name, blocks in ks.items():
            m = re.search(r"attn_rows_kernelILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
            if not m:
                bad.append(f"{name}: template arguments not recognised"); continue
            D, G, paged, direct, U, waves, nt, ub, fuse, shm = (int(x) for x in m.groups())
            flat = [i for b in blocks for i in b]
            bars = [k for k, i in enumerate(flat) if i.startswith("s_barrier")]
            fused = ("v_fma_f32", "v_fmac_f32", "v_pk_fma_f32", "v_fmamk_f32", "v_fmaak_f32")
            if not direct and not (shm and waves == 1) and bars:
                end = bars[1] if len(bars) > 1 else len(flat)
                n = sum(1 for i in flat[bars[0]:end] if i.split()[0].startswith(fused))
                if n:
                    bad.append(f"{name}: {n} fused multiply-add(s) in the cross-wave merge")
        for name, blocks in kernels(text, "attn_merge_kernel").items():
            n = sum(1 for b in blocks for i in b if i.split()[0].startswith(("v_add_f32", "v_pk_add_f32")))
            if n:
                bad.append(f"{name}: {n} separate add(s) in the partition merge")
        if verbose:
            print(f"attention.hip {' '.join(flags) or 'fp16'}: {len(ks)} attn_rows_kernel instantiations checked")
    return bad


def _vregs(tok):
    """VGPR numbers named by one operand token (v7, v[8:9])"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()


def check_gemm_tiled(verbose=True):
    """the TEPI_ROPE epilogue's positions are requested by an inline-asm global_load_dwordx2 in front of the K loop and are only
    complete after the counted wait of K-step 0 — the compiler believes the register pair is defined as soon as the asm has run.  Asserted here,
    per instantiation that holds such a request: no spills / scratch (a spill of the pair would copy it before it has landed), and no
    instruction reads or writes the destination pair between the request and the first s_waitcnt vmcnt(c) that covers it (at least c
    vector-memory loads issued after it: vmcnt retires in issue order)."""
    bad = []
    for build in ((), ("-DNVR_BF16",)):                                  # the fp16 and the bf16 build of the file (Makefile)
        bad += _check_gemm_tiled_build(compile_isa("kernels/gemm_tiled.hip", flags=("-ffp-contract=off",) + build), verbose)
    return bad


def _check_gemm_tiled_build(text, verbose):
    bad = []
    meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
        r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)}
    found = 0
    for m in re.finditer(r"^(_Z\w*gemm_tiled_kernel\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        lines = [l.strip() for l in body.split("\n")]
        reqs = [i for i, l in enumerate(lines) if l.startswith("global_load_dwordx2") and i > 0 and lines[i - 1].startswith(";;#ASMSTART")]
        if not reqs:
            continue
        found += 1
        problems = []
        if any(l.startswith("scratch_") for l in lines):
            problems.append("scratch traffic")
        if name in meta and (meta[name][0] or meta[name][1]):
            problems.append(f"spills {meta[name]}")
        for i in reqs:
            dst = _vregs(lines[i].split()[1].rstrip(","))
            younger, covered = 0, False
            for l in lines[i + 1:]:
                if not l or l.startswith((";", ".")):
                    continue
                w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
                if w and int(w.group(1)) <= younger:
                    covered = True
                    break
                ops = re.findall(r"v\[\d+:\d+\]|v\d+", l)
                if any(_vregs(o) & dst for o in ops):
                    problems.append(f"v{sorted(dst)} touched before its wait: {l}")
                    break
                if l.startswith(("global_load", "buffer_load", "flat_load")):
                    younger += 1
            if not covered and not problems:
                problems.append("no covering wait found")
        if verbose:
            print(f"{name[:90]:90s} {'OK' if not problems else '; '.join(sorted(set(problems)))}  ({len(reqs)} asm position requests)")
        bad += [name + ": " + q for q in sorted(set(problems))]
    if found < 2:
        bad.append(f"only {found} gemm_tiled_kernel instantiations with an inline-asm position request found")
    return bad


def request_problems(lines, i):
    """lines[i] is a vector-memory request issued by inline asm (destination VGPRs = its first operand).  Walks EVERY control-flow path from it (both outcomes of
    each conditional branch, the target of each unconditional one) until a s_waitcnt vmcnt(c) that covers the request (at least c vector-memory loads issued
    behind it on that path: vmcnt retires in issue order) and reports a destination register read or written before that wait, or a path that reaches the end of
    the kernel without one.  The walk knows nothing about which branch outcomes are correlated: code that wants to pass it puts the covering wait on every path."""
    labels = {l[:-1]: k for k, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:$", l)}
    dst = _vregs(lines[i].split()[1].rstrip(","))
    problems, work, seen, uncovered = [], [(i + 1, 0)], {}, False
    while work and not problems:
        k, younger = work.pop()
        while k < len(lines):
            if seen.get(k, 1 << 30) <= younger:                  # (reached before with no more requests behind the one checked: nothing new)
                break
            seen[k] = younger
            l = lines[k]; k += 1
            if not l or l.startswith((";", ".")):
                continue
            w = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if w and int(w.group(1)) <= younger:
                break                                            # covered on this path
            br = re.match(r"(s_branch|s_cbranch_\w+)\s+(\.LBB\d+_\d+)", l)
            if br:
                if br.group(2) in labels:
                    work.append((labels[br.group(2)], younger))
                if br.group(1) == "s_branch":
                    break
                continue
            if l.startswith("s_endpgm"):
                uncovered = True
                break
            ops = re.findall(r"v\[\d+:\d+\]|v\d+", l)
            if any(_vregs(o) & dst for o in ops):
                problems.append(f"v{sorted(dst)} touched before its wait: {l}")
                break
            if l.startswith(("global_load", "buffer_load", "flat_load")):
                younger += 1
        else:
            uncovered = True
    if uncovered and not problems:
        problems.append("a path from the request reaches the end of the kernel without a covering wait")
    return problems


def check_linear_stream(verbose=True):
    """(r06: also the weight pieces of the IMG instantiations — global_load_dwordx4 by inline asm, two chunks in flight behind hand-counted waits.)
    ADVICE r05: linear_stream_kernel<.., SEPI_ROPE, ..> asks for its rows' RoPE positions (global_load_dwordx2) and cache slots (global_load_dword) by
    inline asm in front of the weight stream and ties the registers to an explicit s_waitcnt vmcnt(0) in the epilogue; until then the compiler believes they
    are defined.  Per instantiation holding such requests, both builds: no scratch traffic, no spilled VGPRs (SGPR spills to VGPR lanes are register moves
    and are reported, not failed: the PRE = 2 instantiations carry 4-6), and no instruction reads or writes a destination register between its request and
    the first vmcnt wait that covers it."""
    bad = []
    for build in ((), ("-DNVR_BF16",)):
        text = compile_isa("kernels/linear_stream.hip", flags=("-ffp-contract=off",) + build)
        meta = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in re.finditer(
            r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.sgpr_spill_count:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text)}
        found = 0
        for m in re.finditer(r"^(_Z\w*linear_stream_kernel\w*):[^\n]*\n(.*?)s_endpgm", text, re.S | re.M):
            name, body = m.group(1), m.group(2)
            lines = [l.strip() for l in body.split("\n")]
            reqs = [i for i, l in enumerate(lines) if l.startswith(("global_load_dwordx2", "global_load_dword ", "global_load_dwordx4")) and i > 0 and lines[i - 1].startswith(";;#ASMSTART")]
            if not reqs:
                continue
            found += 1
            problems = []
            if any(l.startswith("scratch_") for l in lines):
                problems.append("scratch traffic")
            if name in meta and meta[name][1]:
                problems.append(f"{meta[name][1]} spilled VGPRs")
            for i in reqs:
                problems += request_problems(lines, i)
            if verbose:
                sg = meta.get(name, (0, 0))[0]
                print(f"{name[:100]:100s} {'OK' if not problems else '; '.join(sorted(set(problems)))}  ({len(reqs)} asm requests, {sg} SGPR spills)")
            bad += [name + ": " + q for q in sorted(set(problems))]
        if found < 2:
            bad.append(f"only {found} linear_stream_kernel instantiations with inline-asm requests found ({' '.join(build) or 'fp16'})")
    return bad


if __name__ == "__main__":
    problems = check_flash() + check_gemm_tiled() + check_linear_stream() + check_attention_merge_forms()
    for p in problems:
        print("FAIL", p)
    sys.exit(1 if problems else 0)
