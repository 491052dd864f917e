"""Control plane of a multi-rank run: a ~100-line TCP rendezvous on MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE.

The data path of tensor-parallel ranks never goes through the host (one-shot peer-to-peer kernels over xGMI, RCCL for large
messages: csrc/comm.cpp).  What the ranks exchange on the host is tiny and rare — the 128-byte RCCL unique id, the 64-byte
hipIpc handles of the peer-to-peer arenas, yes/no agreements, a barrier around the timed region, the maximum of the elapsed
times — so it needs no collective library: a rank process then holds ONE ROCm stack (libnvr.so's), not a second one bundled
with a framework, and exits normally.

Star topology: rank 0 listens, every other rank connects (retrying until `timeout`), every operation is one length-prefixed
message up and one down.  All operations are collective and must be called by every rank in the same order.
"""
import os
import pickle
import socket
import struct
import time
from typing import Any, List, Optional


class SocketGroup:
    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, addr: Optional[str] = None,
                 port: Optional[int] = None, timeout: float = 120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        # (the launcher's own store listens on MASTER_PORT itself: the group takes a port next to it)
        port = int(os.environ.get("MASTER_PORT", "29500")) + 101 if port is None else port
        self.timeout = timeout
        self.peers: List[socket.socket] = []          # rank 0: connection of rank i at index i - 1
        self.up: Optional[socket.socket] = None       # other ranks: the connection to rank 0
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
            srv.listen(self.world)
            srv.settimeout(timeout)
            by_rank = {}
            while len(by_rank) < self.world - 1:
                c, _ = srv.accept()                     # socket.timeout (an OSError) if a rank never shows up
                c.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                c.settimeout(timeout)
                r = struct.unpack("<i", self._recvn(c, 4))[0]
                if not (0 < r < self.world) or r in by_rank:
                    raise RuntimeError(f"rendezvous: unexpected rank {r}")
                by_rank[r] = c
            srv.close()
            self.peers = [by_rank[r] for r in range(1, self.world)]
        else:
            deadline = time.monotonic() + timeout
            while True:
                try:
                    s = socket.create_connection((addr, port), timeout=5.0)
                    break
                except OSError:
                    if time.monotonic() > deadline:
                        raise
                    time.sleep(0.05)
            s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            s.settimeout(timeout)
            s.sendall(struct.pack("<i", self.rank))
            self.up = s

    # ---- framing
    @staticmethod
    def _recvn(s: socket.socket, n: int) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            part = s.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous: peer closed the connection")
            buf += part
        return bytes(buf)

    def _send(self, s: socket.socket, payload: bytes) -> None:
        s.sendall(struct.pack("<q", len(payload)) + payload)

    def _recv(self, s: socket.socket) -> bytes:
        return self._recvn(s, struct.unpack("<q", self._recvn(s, 8))[0])

    # ---- collectives
    def all_gather(self, obj: Any) -> List[Any]:
        """Every rank's object, in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [pickle.loads(self._recv(c)) for c in self.peers]
            blob = pickle.dumps(objs)
            for c in self.peers:
                self._send(c, blob)
            return objs
        self._send(self.up, pickle.dumps(obj))
        return pickle.loads(self._recv(self.up))

    def broadcast(self, obj: Any, root: int = 0) -> Any:
        return self.all_gather(obj if self.rank == root else None)[root]

    def barrier(self) -> None:
        self.all_gather(None)

    def max(self, x: float) -> float:
        return max(self.all_gather(x))

    def min(self, x):
        return min(self.all_gather(x))

    def all_ok(self, flag: bool) -> bool:
        return bool(self.min(1 if flag else 0))

    def close(self) -> None:
        for c in self.peers:
            c.close()
        if self.up is not None:
            self.up.close()
        self.peers, self.up = [], None
