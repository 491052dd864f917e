"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol include/nvr.h declares,
and the product's C++ Sequence / BlockManager / Scheduler are bit-exact with the oracle's literal
restatement of the reference (block ids, block tables, cached-token counts, batch composition,
preemption order, statistics) — on the reference's own unit-test scenarios (file:line cited) and on
randomised traces with prefix sharing, block pressure and preemption."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import nvr_import
from oracle import engine_oracle as eo

nvr = nvr_import.load()


# ---------------------------------------------------------------------------- ABI surface
def test_library_loads_and_exports_every_declared_symbol():
    l = nvr.lib()
    hdr = open(nvr.HEADER_PATH).read()
    names = re.findall(r"NVR_API\s+[^;(]*?\b(nvr_\w+)\s*\(", hdr)
    assert len(names) > 100
    for n in names:
        assert hasattr(l, n), f"{n} declared in include/nvr.h but not exported by libnvr.so"
        assert n in nvr._SIGS, f"{n} has no ctypes binding"
    assert b"gfx950" in l.nvr_version()


def test_error_string_and_status_codes():
    bm = nvr.BlockManager(2, 4)
    s = nvr.Sequence(list(range(12)), nvr.SamplingParams(), 4)
    assert not bm.can_allocate(s)
    with pytest.raises(nvr.NvrError) as e:            # block_manager.rs:531-539
        bm.allocate(s)
    assert e.value.code == -1 and "Not enough free blocks" in str(e.value)
    s2 = nvr.Sequence([1, 2, 3], nvr.SamplingParams(), 4)
    bm.allocate(s2)
    with pytest.raises(nvr.NvrError) as e:            # block_manager.rs:158-160
        bm.allocate(s2)
    assert e.value.code == -2
    s3 = nvr.Sequence([1, 2, 3], nvr.SamplingParams(), 4)
    with pytest.raises(nvr.NvrError) as e:            # block_manager.rs:266-268
        bm.may_append(s3)
    assert e.value.code == -3
    with pytest.raises(nvr.NvrError):                 # block_manager.rs:92 assert -> NVR_ERR_INVARIANT
        nvr.BlockManager(0, 4)


def test_config_and_params_validation():             # config.rs:194-217, sampling_params.rs:127-168
    c = nvr.Config()
    assert (c.c.max_num_batched_tokens, c.c.max_num_seqs, c.c.tensor_parallel_size) == (32768, 512, 1)
    c.validate()
    with pytest.raises(nvr.NvrError):
        nvr.Config(kvcache_block_size=100).validate()
    with pytest.raises(nvr.NvrError):
        nvr.Config(tensor_parallel_size=10).validate()
    # device / dtype strings (config.rs:48-51, validated like :108-116 plus "hip"); same verdicts as the oracle's Config
    assert (c.c.device, c.c.dtype) == (b"hip", b"float16")
    for dev in ("hip", "cuda", "cpu", "metal"):
        nvr.Config(device=dev).validate()
    for dt in ("float16", "bfloat16", "float32"):
        nvr.Config(dtype=dt).validate()
        eo.Config(dtype=dt).validate()
    with pytest.raises(nvr.NvrError, match="Unsupported device: tpu"):
        nvr.Config(device="tpu").validate()
    with pytest.raises(ValueError, match="Unsupported device: tpu"):
        eo.Config(device="tpu").validate()
    with pytest.raises(nvr.NvrError, match="Unsupported dtype: int8"):
        nvr.Config(dtype="int8").validate()
    for chain in (4, 5, 7):                           # (4 / 5 selected launch chains that left the library in r05: scratch/r04_decode_chain/)
        with pytest.raises(nvr.NvrError):
            nvr.Config(decode_chain=chain).validate()
    # the library's own default (the test suite's Config wrapper, conftest.py, is synchronous): launch-ahead is on since r05
    raw = nvr.ConfigC(); nvr.lib().nvr_config_default(C.byref(raw))
    assert raw.async_decode == 1 and raw.decode_chain == 0
    for bad in (dict(temperature=-1.0), dict(max_tokens=0), dict(top_p=1.5), dict(top_k=0)):
        with pytest.raises(nvr.NvrError):
            nvr.SamplingParams(**bad).validate()
    nvr.SamplingParams().validate()
    m = nvr.ModelConfig("qwen3-0.6b")
    assert (m.c.hidden_size, m.c.num_hidden_layers, m.head_dim()) == (1024, 28, 128)
    m.validate(8)
    with pytest.raises(nvr.NvrError):
        m.validate(3)


# ---------------------------------------------------------------------------- reference KATs via the ABI
@pytest.mark.parametrize("toks,prefix,expect", [
    ([1, 2, 3, 4, 5], None, 0xBC50EBD6BC8FA148), ([1, 2, 3, 4, 5], 12345, 0xABC6D5998E62B562),
    ([9, 10, 11, 12], 0x73F859A04F669E6D, 0x8FCDFAB13E7748AA), (list(range(256)), None, 0x486ADFCC62236EFE),
    (list(range(256, 512)), 0x486ADFCC62236EFE, 0x13BD65FA35AB695D), ([], None, 0xEF46DB3751D8E999),
])
def test_compute_hash_vectors(toks, prefix, expect):
    assert nvr.BlockManager.compute_hash(toks, prefix) == expect


def test_compute_hash_matches_oracle_random():
    rng = np.random.default_rng(0)
    for n in [1, 2, 3, 4, 5, 7, 8, 15, 16, 31, 255, 256, 257]:
        t = rng.integers(-2**62, 2**62, n).tolist()
        for pre in (None, int(rng.integers(0, 2**63))):
            assert nvr.BlockManager.compute_hash(t, pre) == eo.BlockManager.compute_hash(t, pre)


def test_bm_reference_scenarios():                   # block_manager.rs:426-506
    bm = nvr.BlockManager(10, 4)
    s = nvr.Sequence(list(range(1, 10)), nvr.SamplingParams(), 4)
    bm.allocate(s)
    assert s.block_table == [0, 1, 2] and bm.get_stats()["free_blocks"] == 7
    bm.deallocate(s)
    assert s.block_table == [] and s.num_cached_tokens == 0 and bm.free_list() == [3, 4, 5, 6, 7, 8, 9, 2, 1, 0]
    bm = nvr.BlockManager(10, 4)
    s1 = nvr.Sequence([1, 2, 3, 4, 5, 6, 7, 8], nvr.SamplingParams(), 4)
    s2 = nvr.Sequence([1, 2, 3, 4, 9, 10, 11, 12], nvr.SamplingParams(), 4)
    bm.allocate(s1); bm.allocate(s2)
    assert s2.num_cached_tokens == 4 and bm.get_block(s1.block_table[0])["ref_count"] == 2
    assert (s1.block_table, s2.block_table) == ([0, 1], [0, 2])
    bm = nvr.BlockManager(10, 4)
    s = nvr.Sequence([1, 2, 3], nvr.SamplingParams(), 4)
    bm.allocate(s)
    s.append_token(4); assert bm.can_append(s); bm.may_append(s); assert len(s.block_table) == 1
    assert bm.get_block(s.block_table[0])["hash"] == 0x73F859A04F669E6D
    s.append_token(5); assert bm.can_append(s); bm.may_append(s); assert len(s.block_table) == 2


def test_sequence_kats():                            # sequence.rs:288-362
    s = nvr.Sequence(list(range(300)), nvr.SamplingParams())
    assert s.num_blocks() == 2 and s.last_block_num_tokens() == 44
    s = nvr.Sequence([1, 2, 3], nvr.SamplingParams(max_tokens=2))
    assert not s.should_stop(None)
    s.append_token(4); assert not s.should_stop(None)
    s.append_token(5); assert s.should_stop(None)
    s = nvr.Sequence([1, 2, 3], nvr.SamplingParams(max_tokens=10)); s.append_token(2); assert s.should_stop(2)
    s = nvr.Sequence([1, 2, 3], nvr.SamplingParams(max_tokens=10, ignore_eos=True)); s.append_token(2)
    assert not s.should_stop(2)
    assert s.status == nvr.WAITING and s.last_token == 2 and s.num_prompt_tokens == 3


def _cfg(**kw):                                      # scheduler.rs:372-387
    d = dict(max_num_seqs=10, max_num_batched_tokens=1000, eos_token_id=2, kvcache_block_size=16,
             num_kvcache_blocks=100, skip_block_size_check=1)
    d.update(kw)
    return d


def test_scheduler_reference_scenarios():            # scheduler.rs:417-578
    sc = nvr.Scheduler(nvr.Config(**_cfg()))
    for p in ([1, 2, 3, 4, 5], [6, 7, 8], [9, 10, 11, 12]):
        sc.add_sequence(nvr.Sequence(p, nvr.SamplingParams(max_tokens=10), 16))
    seqs, pf = sc.schedule()
    assert pf and len(seqs) == 3 and sc.get_queue_lengths() == (0, 3)
    sc = nvr.Scheduler(nvr.Config(**_cfg()))
    sc.add_sequence(nvr.Sequence([1, 2, 3, 4, 5], nvr.SamplingParams(max_tokens=10), 16))
    seqs, pf = sc.schedule(); assert pf
    sc.postprocess(seqs, [6])
    seqs, pf = sc.schedule(); assert not pf and len(seqs) == 1 and len(seqs[0]) == 6
    sc = nvr.Scheduler(nvr.Config(**_cfg(max_num_seqs=2, max_num_batched_tokens=10)))
    for _ in range(5):
        sc.add_sequence(nvr.Sequence([1, 2, 3, 4, 5, 6], nvr.SamplingParams(max_tokens=10), 16))
    seqs, pf = sc.schedule()
    assert pf and len(seqs) == 1 and sc.get_queue_lengths() == (4, 1)
    sc = nvr.Scheduler(nvr.Config(**_cfg()))
    for p in ([1, 2, 3], [4, 5, 6], [7, 8, 9]):
        sc.add_sequence(nvr.Sequence(p, nvr.SamplingParams(max_tokens=1), 16))
    seqs, _ = sc.schedule()
    sc.postprocess(seqs, [10, 11, 12])
    st = sc.get_stats()
    assert (st["total_sequences"], st["finished_sequences"], st["prefill_batches"]) == (3, 3, 1)
    assert st["avg_prefill_batch_size"] == 3.0 and sc.is_finished()
    fin = sc.take_finished()
    assert [f.token_ids[-1] for f in fin] == [10, 11, 12] and all(f.status == nvr.FINISHED for f in fin)
    with pytest.raises(nvr.NvrError) as e:            # scheduler.rs:235-237
        sc.postprocess([], [1])
    assert e.value.code == -4
    with pytest.raises(nvr.NvrError) as e:            # scheduler.rs:218-220
        nvr.Scheduler(nvr.Config(**_cfg())).schedule()
    assert e.value.code == -5


# ---------------------------------------------------------------------------- randomised trace parity
def _snapshot_oracle(sc: eo.Scheduler, seqs, pf):
    st = sc.stats
    return dict(pf=pf, ids=[s.seq_id for s in seqs], tables=[list(s.block_table) for s in seqs],
                chunks=[(s.chunk_start, s.chunk_len) for s in seqs],
                cached=[s.num_cached_tokens for s in seqs], lens=[len(s) for s in seqs],
                free=list(sc.block_manager.free_block_ids), bm=sc.block_manager.get_stats(),
                q=sc.get_queue_lengths(),
                stats=(st.total_sequences, st.finished_sequences, st.preemptions, st.prefill_batches,
                       st.decode_batches, st.avg_prefill_batch_size, st.avg_decode_batch_size))


def _snapshot_product(sc, seqs, pf):
    st = sc.get_stats()
    return dict(pf=pf, ids=[s.seq_id for s in seqs], tables=[s.block_table for s in seqs],
                chunks=[s.chunk for s in seqs],
                cached=[s.num_cached_tokens for s in seqs], lens=[len(s) for s in seqs],
                free=sc.block_manager.free_list(), bm=sc.block_manager.get_stats(), q=sc.get_queue_lengths(),
                stats=(st["total_sequences"], st["finished_sequences"], st["preemptions"], st["prefill_batches"],
                       st["decode_batches"], st["avg_prefill_batch_size"], st["avg_decode_batch_size"]))


@pytest.mark.parametrize("chunked", [False, True])
@pytest.mark.parametrize("seed,bs,nblocks,nseq,max_seqs,budget", [
    (0, 4, 24, 12, 6, 64), (1, 4, 12, 10, 8, 40), (2, 16, 40, 16, 16, 256), (3, 8, 9, 6, 3, 1000),
    (4, 4, 64, 20, 5, 30), (5, 256, 12, 6, 6, 4096), (6, 2, 30, 25, 7, 20), (7, 4, 40, 14, 6, 9), (8, 8, 30, 10, 4, 13),
])
def test_scheduler_trace_parity(seed, bs, nblocks, nseq, max_seqs, budget, chunked):
    """Drive oracle and product with the same requests and the same 'sampled' tokens; every step must
    agree on batch, block tables, free-list order and statistics (covers prefix hits, mid-stream
    arrivals, block exhaustion, victim preemption from running / scheduled / self, EOS and max_tokens)."""
    rng = np.random.default_rng(seed)
    eo.reset_sequence_counter()
    nvr.lib().nvr_seq_reset_id_counter()
    cfg = dict(max_num_seqs=max_seqs, max_num_batched_tokens=budget, eos_token_id=7, kvcache_block_size=bs,
               num_kvcache_blocks=nblocks, enable_chunked_prefill=chunked)
    o = eo.Scheduler(eo.Config(**cfg))
    p = nvr.Scheduler(nvr.Config(skip_block_size_check=1, **cfg))
    shared = rng.integers(8, 50, 3 * bs).tolist()
    pending = []
    for i in range(nseq):
        # chunked (extension A-23): prompts may exceed the token budget and are cut into several prefill steps
        plen = int(rng.integers(1, 5 * bs if chunked else min(budget, 5 * bs)))
        if rng.random() < 0.5:
            k = int(rng.integers(0, 3)) * bs
            prompt = (shared[:k] + rng.integers(8, 50, max(1, plen)).tolist())[:max(1, plen)]
        else:
            prompt = rng.integers(8, 50, plen).tolist()
        sp = dict(max_tokens=int(rng.integers(1, 3 * bs)), ignore_eos=bool(rng.random() < 0.3))
        pending.append((prompt, sp, int(rng.integers(0, 6))))      # arrival step
    step, steps_done, partial_steps = 0, 0, 0
    while True:
        for prompt, sp, arrive in [x for x in pending if x[2] == step]:
            o.add_sequence(eo.Sequence(prompt, eo.SamplingParams(**sp), bs))
            p.add_sequence(nvr.Sequence(prompt, nvr.SamplingParams(**sp), bs))
        pending = [x for x in pending if x[2] > step]
        step += 1
        if o.is_finished():
            assert p.is_finished()
            if not pending:
                break
            continue
        try:
            oseqs, opf = o.schedule()
        except RuntimeError:
            with pytest.raises(nvr.NvrError):
                p.schedule()
            break
        pseqs, ppf = p.schedule()
        assert _snapshot_oracle(o, oseqs, opf) == _snapshot_product(p, pseqs, ppf), f"step {step}"
        partial_steps += int(any(s.chunk_start + s.chunk_len < len(s) for s in oseqs))
        toks = [int((s.seq_id * 131 + len(s) * 17 + seed) % 43) + 5 for s in oseqs]   # 7 == EOS sometimes
        o.postprocess(oseqs, toks)
        p.postprocess(pseqs, toks)
        assert o.get_queue_lengths() == p.get_queue_lengths()
        assert o.block_manager.get_stats() == p.block_manager.get_stats()
        steps_done += 1
        assert steps_done < 5000
    assert steps_done > 3
    assert partial_steps == 0 if not chunked else (partial_steps > 0 or budget >= 5 * bs)   # chunked traces do cut prompts
    ost, pst = o.stats, p.get_stats()
    assert ost.finished_sequences == pst["finished_sequences"] and ost.preemptions == pst["preemptions"]
    fin = p.take_finished()
    assert len(fin) == pst["finished_sequences"]


def test_block_manager_trace_parity_random_ops():
    rng = np.random.default_rng(11)
    bs, nb = 4, 20
    o, p = eo.BlockManager(nb, bs), nvr.BlockManager(nb, bs)
    live = []
    base = rng.integers(0, 5, 4 * bs).tolist()
    for it in range(600):
        r = rng.random()
        if r < 0.35 or not live:
            n = int(rng.integers(1, 4 * bs))
            toks = (base[:int(rng.integers(0, 4)) * bs] + rng.integers(0, 5, n).tolist())[:n]
            so, sp = eo.Sequence(toks, eo.SamplingParams(), bs, seq_id=it), nvr.Sequence(toks, nvr.SamplingParams(), bs)
            assert o.can_allocate(so) == p.can_allocate(sp)
            if o.can_allocate(so):
                o.allocate(so); p.allocate(sp)
                live.append((so, sp))
        elif r < 0.75:
            so, sp = live[int(rng.integers(0, len(live)))]
            t = int(rng.integers(0, 5))
            so.append_token(t); sp.append_token(t)
            assert o.can_append(so) == p.can_append(sp)
            if o.can_append(so):
                o.may_append(so); p.may_append(sp)
            else:                                   # undo: the scheduler would preempt instead
                o.deallocate(so); p.deallocate(sp)
                live.remove((so, sp))
        else:
            so, sp = live.pop(int(rng.integers(0, len(live))))
            o.deallocate(so); p.deallocate(sp)
        for so, sp in live:
            assert so.block_table == sp.block_table and so.num_cached_tokens == sp.num_cached_tokens
        assert list(o.free_block_ids) == p.free_list()
        assert o.get_stats() == p.get_stats()
        for b in range(nb):
            pb = p.get_block(b)
            assert (o.blocks[b].ref_count, o.blocks[b].hash) == (pb["ref_count"], pb["hash"])


def test_trace_scenarios_do_exercise_preemption_and_prefix_hits():
    """Guards the guard: the randomised traces above are only meaningful if they hit the hard paths."""
    eo.reset_sequence_counter()
    o = eo.Scheduler(eo.Config(max_num_seqs=8, max_num_batched_tokens=40, eos_token_id=7, kvcache_block_size=4,
                               num_kvcache_blocks=12))
    rng = np.random.default_rng(1)
    shared = rng.integers(8, 50, 12).tolist()
    for i in range(10):
        o.add_sequence(eo.Sequence(shared[:8] + rng.integers(8, 50, 6).tolist(),
                                   eo.SamplingParams(max_tokens=12, ignore_eos=True), 4))
    cached = 0
    while not o.is_finished():
        seqs, pf = o.schedule()
        cached += sum(s.num_cached_tokens for s in seqs) if pf else 0
        o.postprocess(seqs, [9] * len(seqs))
    assert o.stats.preemptions > 0 and cached > 0


def test_rust_ffi_is_in_sync_with_the_header():
    """integration/rust/src/ffi.rs (the extern "C" block for the reference crate) is generated from include/nvr.h: it must be
    the generator's current output, declare every symbol the header declares, and the hand-written wrappers (hip.rs) may only
    call functions it declares.  (No Rust toolchain in the image: the files are checked as text.)"""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_rust_ffi.py"), "--check"])
    ffi = open(os.path.join(root, "integration", "rust", "src", "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (nvr_\w+)\(", ffi))
    header = set(re.findall(r"NVR_API[^;(]*?\b(nvr_\w+)\s*\(", open(os.path.join(root, "include", "nvr.h")).read()))
    assert declared == header and len(declared) > 140
    used = set(re.findall(r"ffi::(nvr_\w+)\(", open(os.path.join(root, "integration", "rust", "src", "hip.rs")).read()))
    assert used and used <= declared, used - declared


def test_postprocess_appends_for_sequences_the_scheduler_did_not_stamp():
    """scheduler.rs:234-257 appends one token per sequence, whatever called it.  The chunked-prefill extension (A-23) skips the
    append only for a prompt THIS scheduler cut (chunking on, not yet running); a sequence handed to postprocess without having
    been stamped by schedule() (chunk fields zero or stale) still gets its token."""
    for chunked in (0, 1):
        sc = nvr.Scheduler(nvr.Config(max_num_seqs=4, max_num_batched_tokens=64, kvcache_block_size=4, num_kvcache_blocks=16,
                                      skip_block_size_check=1, enable_chunked_prefill=chunked))
        a = nvr.Sequence([1, 2, 3, 4, 5], nvr.SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True), 4)
        sc.add_sequence(a)
        seqs, pf = sc.schedule()
        assert pf and len(seqs) == 1
        sc.postprocess(seqs, [7])
        assert len(seqs[0]) == 6 and seqs[0].token_ids[-1] == 7
        # a stranger: never scheduled here, chunk_len == 0 (the fallback of update_running_sequence, scheduler.rs:272-273)
        b = nvr.Sequence([9, 9, 9], nvr.SamplingParams(temperature=0.0, max_tokens=4, ignore_eos=True), 4)
        assert tuple(b.chunk) == (0, 0)
        sc.postprocess([b], [5])                 # joins the running queue (scheduler.rs:272-273): the scheduler owns it from here on
        b.owned = False
        assert len(b) == 4 and b.token_ids[-1] == 5


def test_schedule_capacity_is_bounded_by_live_sequences():
    """nvr_sched_schedule refuses an output array that cannot hold the batch BEFORE anything moves; the bound is
    min(max_num_seqs, live sequences), so a caller with two requests does not need a 512-entry array."""
    import ctypes as C
    l = nvr.lib()
    sc = nvr.Scheduler(nvr.Config(max_num_seqs=512, max_num_batched_tokens=64, kvcache_block_size=4, num_kvcache_blocks=16, skip_block_size_check=1))
    for p in ([1, 2, 3], [4, 5]):
        sc.add_sequence(nvr.Sequence(p, nvr.SamplingParams(temperature=0.0, max_tokens=2, ignore_eos=True), 4))
    out = (C.c_void_p * 2)(); n = C.c_size_t(); pf = C.c_int()
    assert l.nvr_sched_schedule(sc.h, out, 1, C.byref(n), C.byref(pf)) == -7            # NVR_ERR_INVALID_ARG, nothing scheduled
    assert sc.get_queue_lengths() == (2, 0)
    nvr.check(l.nvr_sched_schedule(sc.h, out, 2, C.byref(n), C.byref(pf)))
    assert n.value == 2 and pf.value == 1


def test_activation_type_from_str():
    """ActivationType::from_str, src/layers/activation.rs:169-182 (case-insensitive, the aliases, the error text) through the C ABI (no GPU needed)."""
    import ctypes as C
    l = nvr.lib()
    want = {"silu": 0, "swish": 0, "SiLU": 0, "gelu": 1, "GELU": 1, "relu": 2, "silu_and_mul": 3, "SiluAndMul": 3, "gelu_and_mul": 4, "geluandmul": 4}
    for name, kind in want.items():
        out = C.c_int32(-1)
        assert l.nvr_activation_type_from_str(name.encode(), C.byref(out)) == 0 and out.value == kind, name
    out = C.c_int32(-1)
    assert l.nvr_activation_type_from_str(b"tanh", C.byref(out)) == -7            # NVR_ERR_INVALID_ARG
    assert "Unknown activation function: tanh" in nvr.last_error()
