"""Redis-schema-compatible job store (component C8 + external X8).

The reference keeps all job state in Redis (``util/RedisHelper.py:1-17``) and
coordinates its three daemons purely through atomic Redis commands
(SADD/SET ``handlers/train_handler.py:134-135``; HSETNX/HINCRBY/HGET/HDEL/RPUSH
``eventHandlers/CleanJobHandler.py:27-80``; BLPOP ``service/remove_ps.py:58``).
Neither ``redis`` nor ``redis-server`` exists in this image and the GPU box has
no network, so this module provides the whole stack:

* :class:`MemoryStore`  -- thread-safe in-process implementation of the command
  subset (plus the few extra commands the new lifecycle needs: SREM, SMEMBERS,
  HSET, HGETALL, LPUSH, LPOP, LLEN, LRANGE, DEL, EXISTS, KEYS, INCR, PING).
* :class:`StoreServer`  -- RESP2 (Redis wire protocol) TCP server around a
  MemoryStore, so separate daemons (REST, watcher, reaper, ranks) share state
  exactly as they would through a real Redis.  A C++ epoll implementation of
  the same protocol lives in ``native/store_server.cpp``.
* :class:`StoreClient`  -- RESP2 client with redis-py method names; it speaks to
  StoreServer, to the native server, or to a real Redis.
* :class:`RedisHelper`  -- the reference's connection singleton: ``getRedis()``.

All values are returned as ``str`` (``decode_responses=True`` semantics) so the
reference's ``'psCnt: ' + psCnt`` style code keeps working.
"""
from __future__ import annotations

import fnmatch
import socket
import socketserver
import threading
import time
from collections import deque

from .config import ApiConfig


def _s(v):
    if isinstance(v, bytes):
        return v.decode()
    if isinstance(v, bool):
        return "1" if v else "0"
    return str(v)


class WrongType(Exception):
    pass


class MemoryStore:
    """In-process store with Redis semantics for the command subset we use."""

    def __init__(self):
        self._d = {}
        self._cv = threading.Condition()

    # -- helpers ----------------------------------------------------------
    def _typed(self, key, typ, create=True):
        v = self._d.get(key)
        if v is None:
            if not create:
                return None
            v = typ()
            self._d[key] = v
        elif not isinstance(v, typ):
            raise WrongType("WRONGTYPE Operation against a key holding the wrong kind of value")
        return v

    def _gc(self, key):
        v = self._d.get(key)
        if v is not None and not isinstance(v, str) and len(v) == 0:
            del self._d[key]

    # -- strings ----------------------------------------------------------
    def set(self, key, value, nx=False):
        with self._cv:
            key = _s(key)
            if nx and key in self._d:
                return None
            self._d[key] = _s(value)
            return True

    def setnx(self, key, value):
        return 1 if self.set(key, value, nx=True) else 0

    def get(self, key):
        with self._cv:
            v = self._d.get(_s(key))
            if v is not None and not isinstance(v, str):
                raise WrongType("WRONGTYPE")
            return v

    def incr(self, key, amount=1):
        with self._cv:
            key = _s(key)
            v = int(self._d.get(key, "0")) + int(amount)
            self._d[key] = str(v)
            return v

    def delete(self, *keys):
        with self._cv:
            n = 0
            for k in keys:
                if self._d.pop(_s(k), None) is not None:
                    n += 1
            return n

    def exists(self, *keys):
        with self._cv:
            return sum(1 for k in keys if _s(k) in self._d)

    def keys(self, pattern="*"):
        with self._cv:
            return [k for k in self._d if fnmatch.fnmatchcase(k, _s(pattern))]

    def flushall(self):
        with self._cv:
            self._d.clear()
            return True

    def ping(self):
        return True

    # -- sets -------------------------------------------------------------
    def sadd(self, key, *members):
        with self._cv:
            s = self._typed(_s(key), set)
            n = 0
            for m in members:
                m = _s(m)
                if m not in s:
                    s.add(m)
                    n += 1
            return n

    def srem(self, key, *members):
        with self._cv:
            s = self._typed(_s(key), set, create=False)
            if s is None:
                return 0
            n = 0
            for m in members:
                m = _s(m)
                if m in s:
                    s.discard(m)
                    n += 1
            self._gc(_s(key))
            return n

    def smembers(self, key):
        with self._cv:
            s = self._typed(_s(key), set, create=False)
            return set(s) if s else set()

    def sismember(self, key, member):
        with self._cv:
            s = self._typed(_s(key), set, create=False)
            return bool(s) and _s(member) in s

    def scard(self, key):
        with self._cv:
            s = self._typed(_s(key), set, create=False)
            return len(s) if s else 0

    # -- hashes -----------------------------------------------------------
    def hset(self, key, field, value):
        with self._cv:
            h = self._typed(_s(key), dict)
            new = _s(field) not in h
            h[_s(field)] = _s(value)
            return 1 if new else 0

    def hsetnx(self, key, field, value):
        with self._cv:
            h = self._typed(_s(key), dict)
            if _s(field) in h:
                return 0
            h[_s(field)] = _s(value)
            return 1

    def hget(self, key, field):
        with self._cv:
            h = self._typed(_s(key), dict, create=False)
            return None if h is None else h.get(_s(field))

    def hgetall(self, key):
        with self._cv:
            h = self._typed(_s(key), dict, create=False)
            return dict(h) if h else {}

    def hincrby(self, key, field, amount=1):
        with self._cv:
            h = self._typed(_s(key), dict)
            v = int(h.get(_s(field), "0")) + int(amount)
            h[_s(field)] = str(v)
            return v

    def hdel(self, key, *fields):
        with self._cv:
            h = self._typed(_s(key), dict, create=False)
            if h is None:
                return 0
            n = 0
            for f in fields:
                if h.pop(_s(f), None) is not None:
                    n += 1
            self._gc(_s(key))
            return n

    def hexists(self, key, field):
        with self._cv:
            h = self._typed(_s(key), dict, create=False)
            return bool(h) and _s(field) in h

    def hlen(self, key):
        with self._cv:
            h = self._typed(_s(key), dict, create=False)
            return len(h) if h else 0

    # -- lists ------------------------------------------------------------
    def rpush(self, key, *values):
        with self._cv:
            q = self._typed(_s(key), deque)
            for v in values:
                q.append(_s(v))
            self._cv.notify_all()
            return len(q)

    def lpush(self, key, *values):
        with self._cv:
            q = self._typed(_s(key), deque)
            for v in values:
                q.appendleft(_s(v))
            self._cv.notify_all()
            return len(q)

    def lpop(self, key):
        with self._cv:
            q = self._typed(_s(key), deque, create=False)
            if not q:
                return None
            v = q.popleft()
            self._gc(_s(key))
            return v

    def llen(self, key):
        with self._cv:
            q = self._typed(_s(key), deque, create=False)
            return len(q) if q else 0

    def lrange(self, key, start, stop):
        with self._cv:
            q = self._typed(_s(key), deque, create=False)
            if not q:
                return []
            lst = list(q)
            start, stop = int(start), int(stop)
            n = len(lst)
            if start < 0:
                start = max(0, n + start)
            if stop < 0:
                stop = n + stop
            return lst[start:stop + 1]

    def blpop(self, keys, timeout=0):
        """Redis BLPOP: ``timeout`` 0 blocks forever; returns ``(key, value)`` or None."""
        if isinstance(keys, (str, bytes)):
            keys = [keys]
        keys = [_s(k) for k in keys]
        deadline = None if not timeout else time.monotonic() + float(timeout)
        with self._cv:
            while True:
                for k in keys:
                    q = self._typed(k, deque, create=False)
                    if q:
                        v = q.popleft()
                        self._gc(k)
                        return (k, v)
                if deadline is None:
                    self._cv.wait(1.0)
                else:
                    left = deadline - time.monotonic()
                    if left <= 0:
                        return None
                    self._cv.wait(min(left, 1.0))

    def pipeline(self):
        return _Pipeline(self)


class _Pipeline:
    """Minimal MULTI-less pipeline: queues calls, ``execute()`` runs them in
    order (finishes the reference's ``# TODO pipeline`` at
    ``handlers/train_handler.py:133``)."""

    def __init__(self, target):
        self._t = target
        self._calls = []

    def __getattr__(self, name):
        fn = getattr(self._t, name)

        def queue(*a, **kw):
            self._calls.append((fn, a, kw))
            return self

        return queue

    def execute(self):
        calls, self._calls = self._calls, []
        return [fn(*a, **kw) for fn, a, kw in calls]

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


# ---------------------------------------------------------------------------
# RESP2 wire protocol
# ---------------------------------------------------------------------------

def _encode_reply(v):
    if v is None:
        return b"$-1\r\n"
    if v is True:
        return b"+OK\r\n"
    if isinstance(v, int):
        return b":%d\r\n" % v
    if isinstance(v, (str, bytes)):
        b = v.encode() if isinstance(v, str) else v
        return b"$%d\r\n%s\r\n" % (len(b), b)
    if isinstance(v, (list, tuple, set)):
        items = sorted(v) if isinstance(v, set) else list(v)
        return b"*%d\r\n" % len(items) + b"".join(_encode_reply(x) for x in items)
    if isinstance(v, dict):
        flat = []
        for k, x in v.items():
            flat += [k, x]
        return _encode_reply(flat)
    raise TypeError(type(v))


class _Reader:
    def __init__(self, rfile):
        self.f = rfile

    def read(self):
        line = self.f.readline()
        if not line:
            raise ConnectionError("closed")
        t, rest = line[:1], line[1:-2]
        if t == b"+":
            return rest.decode()
        if t == b"-":
            return RuntimeError(rest.decode())
        if t == b":":
            return int(rest)
        if t == b"$":
            n = int(rest)
            if n < 0:
                return None
            data = self.f.read(n + 2)
            return data[:-2].decode()
        if t == b"*":
            n = int(rest)
            if n < 0:
                return None
            return [self.read() for _ in range(n)]
        # inline command (telnet-style)
        return line.decode().split()


_COMMANDS = {
    "PING": lambda st, a: "PONG" if not a else a[0],
    "SET": lambda st, a: st.set(a[0], a[1], nx=any(x.upper() == "NX" for x in a[2:])),
    "SETNX": lambda st, a: st.setnx(a[0], a[1]),
    "GET": lambda st, a: st.get(a[0]),
    "INCR": lambda st, a: st.incr(a[0]),
    "INCRBY": lambda st, a: st.incr(a[0], int(a[1])),
    "DEL": lambda st, a: st.delete(*a),
    "EXISTS": lambda st, a: st.exists(*a),
    "KEYS": lambda st, a: st.keys(a[0]),
    "FLUSHALL": lambda st, a: st.flushall(),
    "SADD": lambda st, a: st.sadd(a[0], *a[1:]),
    "SREM": lambda st, a: st.srem(a[0], *a[1:]),
    "SMEMBERS": lambda st, a: st.smembers(a[0]),
    "SISMEMBER": lambda st, a: int(st.sismember(a[0], a[1])),
    "SCARD": lambda st, a: st.scard(a[0]),
    "HSET": lambda st, a: sum(st.hset(a[0], a[i], a[i + 1]) for i in range(1, len(a) - 1, 2)),
    "HSETNX": lambda st, a: st.hsetnx(a[0], a[1], a[2]),
    "HGET": lambda st, a: st.hget(a[0], a[1]),
    "HGETALL": lambda st, a: st.hgetall(a[0]),
    "HINCRBY": lambda st, a: st.hincrby(a[0], a[1], int(a[2])),
    "HDEL": lambda st, a: st.hdel(a[0], *a[1:]),
    "HEXISTS": lambda st, a: int(st.hexists(a[0], a[1])),
    "HLEN": lambda st, a: st.hlen(a[0]),
    "RPUSH": lambda st, a: st.rpush(a[0], *a[1:]),
    "LPUSH": lambda st, a: st.lpush(a[0], *a[1:]),
    "LPOP": lambda st, a: st.lpop(a[0]),
    "LLEN": lambda st, a: st.llen(a[0]),
    "LRANGE": lambda st, a: st.lrange(a[0], int(a[1]), int(a[2])),
    "BLPOP": lambda st, a: st.blpop(a[:-1], float(a[-1])),
}


class _Handler(socketserver.StreamRequestHandler):
    def handle(self):
        self.request.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        rd = _Reader(self.rfile)
        st = self.server.store
        while True:
            try:
                cmd = rd.read()
            except (ConnectionError, OSError, ValueError):
                return
            if not cmd:
                continue
            name, args = str(cmd[0]).upper(), cmd[1:]
            if name == "QUIT":
                self.wfile.write(b"+OK\r\n")
                return
            fn = _COMMANDS.get(name)
            try:
                if fn is None:
                    out = b"-ERR unknown command '%s'\r\n" % name.encode()
                else:
                    res = fn(st, args)
                    if name == "PING" and res == "PONG":
                        out = b"+PONG\r\n"
                    elif name == "BLPOP" and res is None:
                        out = b"*-1\r\n"
                    else:
                        out = _encode_reply(res)
            except WrongType as e:
                out = b"-%s\r\n" % str(e).encode()
            except Exception as e:  # malformed arguments etc.
                out = b"-ERR %s\r\n" % str(e).replace("\r", " ").replace("\n", " ").encode()
            try:
                self.wfile.write(out)
                self.wfile.flush()
            except OSError:
                return


class _TCPServer(socketserver.ThreadingTCPServer):
    allow_reuse_address = True
    daemon_threads = True


class StoreServer:
    """RESP2 server; ``port=0`` picks an ephemeral port (see ``.port``)."""

    def __init__(self, host="127.0.0.1", port=0, store=None):
        self.store = store or MemoryStore()
        self._srv = _TCPServer((host, port), _Handler)
        self._srv.store = self.store
        self.host, self.port = self._srv.server_address[:2]
        self._thread = None

    def start(self):
        self._thread = threading.Thread(target=self._srv.serve_forever, name="dtf-store", daemon=True)
        self._thread.start()
        return self

    def serve_forever(self):
        self._srv.serve_forever()

    def stop(self):
        self._srv.shutdown()
        self._srv.server_close()

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.stop()
        return False


class StoreClient:
    """RESP2 client exposing redis-py method names (``decode_responses=True``)."""

    def __init__(self, host="127.0.0.1", port=6379, timeout=None):
        self.host, self.port = host, int(port)
        self._timeout = timeout
        self._lock = threading.RLock()
        self._sock = None
        self._rd = None

    def _connect(self):
        s = socket.create_connection((self.host, self.port), timeout=self._timeout or 10)
        s.settimeout(None)
        s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
        self._sock = s
        self._rd = _Reader(s.makefile("rb"))

    def close(self):
        with self._lock:
            if self._sock is not None:
                try:
                    self._sock.close()
                finally:
                    self._sock = None

    @staticmethod
    def _pack(args):
        out = [b"*%d\r\n" % len(args)]
        for a in args:
            b = _s(a).encode()
            out.append(b"$%d\r\n%s\r\n" % (len(b), b))
        return b"".join(out)

    def execute_command(self, *args):
        with self._lock:
            for attempt in (0, 1):
                try:
                    if self._sock is None:
                        self._connect()
                    self._sock.sendall(self._pack(args))
                    r = self._rd.read()
                    break
                except (ConnectionError, OSError):
                    self.close()
                    if attempt:
                        raise
            if isinstance(r, RuntimeError):
                raise r
            return r

    def execute_many(self, commands):
        """One round trip for several commands (pipeline)."""
        with self._lock:
            if self._sock is None:
                self._connect()
            self._sock.sendall(b"".join(self._pack(c) for c in commands))
            out = [self._rd.read() for _ in commands]
        for r in out:
            if isinstance(r, RuntimeError):
                raise r
        return out

    # strings
    def ping(self):
        return self.execute_command("PING") == "PONG"

    def set(self, key, value, nx=False):
        r = self.execute_command(*(["SET", key, value] + (["NX"] if nx else [])))
        return True if r == "OK" else None

    def setnx(self, key, value):
        return self.execute_command("SETNX", key, value)

    def get(self, key):
        return self.execute_command("GET", key)

    def incr(self, key, amount=1):
        return self.execute_command("INCRBY", key, amount)

    def delete(self, *keys):
        return self.execute_command("DEL", *keys)

    def exists(self, *keys):
        return self.execute_command("EXISTS", *keys)

    def keys(self, pattern="*"):
        return self.execute_command("KEYS", pattern)

    def flushall(self):
        return self.execute_command("FLUSHALL") == "OK"

    # sets
    def sadd(self, key, *m):
        return self.execute_command("SADD", key, *m)

    def srem(self, key, *m):
        return self.execute_command("SREM", key, *m)

    def smembers(self, key):
        return set(self.execute_command("SMEMBERS", key) or [])

    def sismember(self, key, m):
        return bool(self.execute_command("SISMEMBER", key, m))

    def scard(self, key):
        return self.execute_command("SCARD", key)

    # hashes
    def hset(self, key, field, value):
        return self.execute_command("HSET", key, field, value)

    def hsetnx(self, key, field, value):
        return self.execute_command("HSETNX", key, field, value)

    def hget(self, key, field):
        return self.execute_command("HGET", key, field)

    def hgetall(self, key):
        flat = self.execute_command("HGETALL", key) or []
        return dict(zip(flat[0::2], flat[1::2]))

    def hincrby(self, key, field, amount=1):
        return self.execute_command("HINCRBY", key, field, amount)

    def hdel(self, key, *fields):
        return self.execute_command("HDEL", key, *fields)

    def hexists(self, key, field):
        return bool(self.execute_command("HEXISTS", key, field))

    def hlen(self, key):
        return self.execute_command("HLEN", key)

    # lists
    def rpush(self, key, *v):
        return self.execute_command("RPUSH", key, *v)

    def lpush(self, key, *v):
        return self.execute_command("LPUSH", key, *v)

    def lpop(self, key):
        return self.execute_command("LPOP", key)

    def llen(self, key):
        return self.execute_command("LLEN", key)

    def lrange(self, key, start, stop):
        return self.execute_command("LRANGE", key, start, stop) or []

    def blpop(self, keys, timeout=0):
        if isinstance(keys, (str, bytes)):
            keys = [keys]
        r = self.execute_command("BLPOP", *(list(keys) + [timeout]))
        return None if r is None else (r[0], r[1])

    def pipeline(self):
        return _ClientPipeline(self)


class _ClientPipeline:
    _NAMES = {"set": "SET", "sadd": "SADD", "srem": "SREM", "hset": "HSET", "hsetnx": "HSETNX",
              "hdel": "HDEL", "hincrby": "HINCRBY", "rpush": "RPUSH", "delete": "DEL",
              "get": "GET", "hget": "HGET"}

    def __init__(self, client):
        self._c = client
        self._cmds = []

    def __getattr__(self, name):
        cmd = self._NAMES[name]

        def queue(*a):
            self._cmds.append((cmd,) + a)
            return self

        return queue

    def execute(self):
        cmds, self._cmds = self._cmds, []
        return self._c.execute_many(cmds) if cmds else []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class RedisHelper:
    """Connection singleton (reference ``util/RedisHelper.py:1-17``).

    ``getRedis()`` returns, in order of preference: an object installed with
    :meth:`install` (all-in-one service and tests share a MemoryStore this way),
    a ``redis.Redis`` when ``redis`` is importable and ``[redis] backend`` allows
    it, else a :class:`StoreClient` to ``[redis] host:port``.
    """

    _instance = None
    _lock = threading.Lock()

    def __new__(cls):
        with cls._lock:
            if cls._instance is None:
                inst = super().__new__(cls)
                inst._conn = None
                inst._local = threading.local()
                cls._instance = inst
        return cls._instance

    @classmethod
    def install(cls, conn):
        cls()._conn = conn

    @classmethod
    def reset(cls):
        inst = cls()
        inst._conn = None
        inst._local = threading.local()

    def getRedis(self):
        if self._conn is not None:
            return self._conn
        c = getattr(self._local, "c", None)
        if c is None:
            cfg = ApiConfig()
            host, port = cfg.get("redis", "host"), cfg.getint("redis", "port")
            backend = cfg.get("redis", "backend", "auto")
            c = None
            if backend in ("auto", "redis"):
                try:
                    import redis  # noqa: F401  (absent in this image)

                    c = redis.Redis(host=host, port=port, decode_responses=True)
                except ImportError:
                    if backend == "redis":
                        raise
            if c is None:
                c = StoreClient(host, port)
            # one connection per thread: BLPOP in the reaper must not block
            # the handlers' commands
            self._local.c = c
        return c


def main(argv=None):
    import argparse

    ap = argparse.ArgumentParser(description="RESP2 job-store server (Redis-compatible subset)")
    ap.add_argument("--host", default=None)
    ap.add_argument("--port", type=int, default=None)
    a = ap.parse_args(argv)
    cfg = ApiConfig()
    srv = StoreServer(a.host or cfg.get("redis", "host"), a.port if a.port is not None else cfg.getint("redis", "port"))
    print("dtf store listening on %s:%d" % (srv.host, srv.port), flush=True)
    try:
        srv.serve_forever()
    except KeyboardInterrupt:
        pass


if __name__ == "__main__":
    main()
