"""nano-vllm-rs_amd/ctrl.py — the TCP rendezvous that carries a multi-rank run's control plane (unique-id broadcast, hipIpc handle exchange,
agreements, barrier, max): exercised here as real processes on 127.0.0.1, CPU only."""
import multiprocessing as mp
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import nvr_import  # noqa: E402


def _free_port() -> int:
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _rank_main(rank, world, port, q):
    try:
        ctrl = nvr_import.load_ctrl()
        g = ctrl.SocketGroup(rank=rank, world=world, addr="127.0.0.1", port=port, timeout=30.0)
        out = {}
        out["uid"] = g.broadcast(bytes(range(128)) if rank == 0 else None, 0)                 # the RCCL unique id: 128 bytes from rank 0
        out["handles"] = g.all_gather((bytes([rank]) * 64, rank % 2))                         # hipIpc handles + device ordinals, rank order
        out["agree_all"] = g.all_ok(True)
        out["agree_one_no"] = g.all_ok(rank != world - 1)                                     # one rank failed locally: nobody goes on
        g.barrier()
        out["max"] = g.max(1.5 + rank)                                                        # elapsed time: the slowest rank's
        out["min"] = g.min(10 - rank)
        out["root2"] = g.broadcast(("from", rank) if rank == world - 1 else None, world - 1)  # any root
        for _ in range(50):                                                                   # many small rounds stay in step
            assert g.max(rank) == world - 1
        g.barrier(); g.close()
        q.put((rank, out))
    except BaseException as ex:                                                               # noqa: BLE001
        q.put((rank, repr(ex)))


@pytest.mark.parametrize("world", [2, 4, 8])
def test_socket_group_collectives(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    ps = [ctx.Process(target=_rank_main, args=(r, world, port, q)) for r in range(world)]
    for p in reversed(ps): p.start()                                                          # rank 0 (the listener) last: the others must retry
    res = dict(q.get(timeout=60) for _ in range(world))
    for p in ps: p.join(30)
    assert all(p.exitcode == 0 for p in ps)
    for r in range(world):
        o = res[r]
        assert isinstance(o, dict), o
        assert o["uid"] == bytes(range(128))
        assert o["handles"] == [(bytes([i]) * 64, i % 2) for i in range(world)]
        assert o["agree_all"] is True and o["agree_one_no"] is False
        assert o["max"] == 1.5 + world - 1 and o["min"] == 10 - (world - 1)
        assert o["root2"] == ("from", world - 1)


def test_socket_group_single_rank_and_missing_peer():
    ctrl = nvr_import.load_ctrl()
    g = ctrl.SocketGroup(rank=0, world=1)
    assert g.all_gather("x") == ["x"] and g.max(2.0) == 2.0 and g.all_ok(True)
    g.barrier(); g.close()
    with pytest.raises(OSError):                                       # rank 1 never connects: rank 0 gives up after its timeout instead of hanging
        ctrl.SocketGroup(rank=0, world=2, addr="127.0.0.1", port=_free_port(), timeout=0.5)
    with pytest.raises(OSError):                                       # no listener: a rank gives up too
        ctrl.SocketGroup(rank=1, world=2, addr="127.0.0.1", port=_free_port(), timeout=0.5)


def test_wire_format_is_plain_data_and_frames_are_authenticated(monkeypatch):
    """nothing that arrives on the control-plane socket is unpickled.  The codec round-trips exactly the values the ranks
    exchange (bytes, tuples, nested lists, None / bool / int / float / str), refuses anything else, and a decoded frame can only ever be
    plain data; with NVR_CTRL_SECRET set a frame whose HMAC does not verify — or an oversized one — is refused."""
    import struct
    ctrl = nvr_import.load_ctrl()
    vals = [None, True, 3, -2.5, "x", bytes(range(256)), (bytes(64), 1), [(b"\x00\xff", 0), (b"", 7)], {"a": [1, (2, 3)], "b": b"zz"},
            ([[1, 2], [3]], [0, 1, True, 5, 2, False])]
    for v in vals:
        assert ctrl.loads(ctrl.dumps(v)) == v and type(ctrl.loads(ctrl.dumps(v))) is type(v)
    import pickle
    for bad in (object(), {1: 2}, {"__x": 1}, {1, 2}):
        with pytest.raises(TypeError):
            ctrl.dumps(bad)
    with pytest.raises(ValueError):                                    # a pickle is not a frame
        ctrl.loads(pickle.dumps({"a": 1}))
    assert "pickle" not in open(ctrl.__file__).read().replace("unpickled", "")

    a, b = socket.socketpair()
    try:
        monkeypatch.setenv("NVR_CTRL_SECRET", "s3cret")
        g = ctrl.SocketGroup(rank=0, world=1)
        g._send(a, ctrl.dumps(("ok", 1)))
        assert ctrl.loads(g._recv(b)) == ("ok", 1)
        monkeypatch.setenv("NVR_CTRL_SECRET", "other")
        g._send(a, ctrl.dumps("forged"))                               # (sender still holds the first key: built before the change)
        h = ctrl.SocketGroup(rank=0, world=1)
        with pytest.raises(ConnectionError, match="does not verify"):
            h._recv(b)
        a.sendall(struct.pack("<q", ctrl.MAX_FRAME + 1))
        with pytest.raises(ConnectionError, match="refused"):
            h._recv(b)
    finally:
        a.close(); b.close()
