"""Control plane of a multi-rank run: a ~100-line TCP rendezvous on MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE.

The data path of tensor-parallel ranks never goes through the host (one-shot peer-to-peer kernels over xGMI, RCCL for large
messages: csrc/comm.cpp).  What the ranks exchange on the host is tiny and rare — the 128-byte RCCL unique id, the 64-byte
hipIpc handles of the peer-to-peer arenas, yes/no agreements, a barrier around the timed region, the maximum of the elapsed
times — so it needs no collective library: a rank process then holds ONE ROCm stack (libnvr.so's), not a second one bundled
with a framework, and exits normally.

Star topology: rank 0 listens, every other rank connects (retrying until `timeout`), every operation is one length-prefixed
message up and one down.  All operations are collective and must be called by every rank in the same order.

Wire format (nothing that arrives on the socket is ever unpickled): a frame is JSON — None, bool, int, float, str, list, dict
with string keys, and two tagged forms, {"__b": base64} for bytes and {"__t": [...]} for tuples — of at most MAX_FRAME bytes; decoding
builds plain data only.  With NVR_CTRL_SECRET set in the ranks' environment (the launcher's job) every frame also carries an HMAC-SHA256 of
its body under that secret and a frame whose tag does not verify is refused; the listener binds MASTER_ADDR (loopback by default), not
every interface.
"""
import base64
import hashlib
import hmac
import json
import os
import socket
import struct
import time
from typing import Any, List, Optional

MAX_FRAME = 64 << 20


def _enc(o: Any) -> Any:
    if o is None or isinstance(o, (bool, int, float, str)):
        return o
    if isinstance(o, (bytes, bytearray)):
        return {"__b": base64.b64encode(bytes(o)).decode("ascii")}
    if isinstance(o, tuple):
        return {"__t": [_enc(x) for x in o]}
    if isinstance(o, list):
        return [_enc(x) for x in o]
    if isinstance(o, dict):
        if not all(isinstance(k, str) and not k.startswith("__") for k in o):
            raise TypeError("control plane: dict keys must be strings that do not start with '__'")
        return {k: _enc(v) for k, v in o.items()}
    if hasattr(o, "item") and getattr(o, "shape", None) == ():       # a numpy scalar
        return _enc(o.item())
    raise TypeError(f"control plane: cannot send a {type(o).__name__} (plain data only)")


def _dec(o: Any) -> Any:
    if isinstance(o, list):
        return [_dec(x) for x in o]
    if isinstance(o, dict):
        if set(o) == {"__b"}:
            return base64.b64decode(o["__b"], validate=True)
        if set(o) == {"__t"}:
            return tuple(_dec(x) for x in o["__t"])
        return {k: _dec(v) for k, v in o.items()}
    return o


def dumps(obj: Any) -> bytes:
    return json.dumps(_enc(obj), separators=(",", ":"), allow_nan=True).encode("utf-8")


def loads(blob: bytes) -> Any:
    return _dec(json.loads(blob.decode("utf-8")))


class SocketGroup:
    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, addr: Optional[str] = None,
                 port: Optional[int] = None, timeout: float = 120.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else world
        addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
        # (the launcher's own store listens on MASTER_PORT itself: the group takes a port next to it)
        port = int(os.environ.get("MASTER_PORT", "29500")) + 101 if port is None else port
        self.timeout = timeout
        secret = os.environ.get("NVR_CTRL_SECRET", "")
        self._key = secret.encode("utf-8") if secret else None
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
        if self._key is not None:
            payload = hmac.new(self._key, payload, hashlib.sha256).digest() + payload
        s.sendall(struct.pack("<q", len(payload)) + payload)

    def _recv(self, s: socket.socket) -> bytes:
        n = struct.unpack("<q", self._recvn(s, 8))[0]
        if n < 0 or n > MAX_FRAME:
            raise ConnectionError(f"rendezvous: frame of {n} bytes refused (limit {MAX_FRAME})")
        payload = self._recvn(s, n)
        if self._key is not None:
            if n < 32 or not hmac.compare_digest(payload[:32], hmac.new(self._key, payload[32:], hashlib.sha256).digest()):
                raise ConnectionError("rendezvous: frame refused (NVR_CTRL_SECRET: the authentication tag does not verify)")
            payload = payload[32:]
        return payload

    # ---- collectives
    def all_gather(self, obj: Any) -> List[Any]:
        """Every rank's object, in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            objs = [obj] + [loads(self._recv(c)) for c in self.peers]
            blob = dumps(objs)
            for c in self.peers:
                self._send(c, blob)
            return objs
        self._send(self.up, dumps(obj))
        return loads(self._recv(self.up))

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
