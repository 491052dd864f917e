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


if __name__ == "__main__":
    problems = check_flash()
    for p in problems:
        print("FAIL", p)
    sys.exit(1 if problems else 0)
