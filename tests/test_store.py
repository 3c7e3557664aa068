"""Job store: the Redis command subset the reference uses, in
process and over the RESP wire."""
import threading
import time

import pytest

from distributed_tensorflow_amd.utils.store import MemoryStore, StoreClient, StoreServer


@pytest.fixture(params=["memory", "resp", "native"])
def rc(request):
    if request.param == "memory":
        yield MemoryStore()
    elif request.param == "native":
        from distributed_tensorflow_amd.native.build import spawn_store

        proc, port = spawn_store()
        c = StoreClient("127.0.0.1", port)
        yield c
        c.close()
        proc.kill()
        proc.wait()
    else:
        with StoreServer("127.0.0.1", 0) as srv:
            c = StoreClient(srv.host, srv.port)
            yield c
            c.close()


def test_strings_and_sets(rc):
    assert rc.set("uid", '{"a": 1}') is True
    assert rc.get("uid") == '{"a": 1}'
    assert rc.get("missing") is None
    assert rc.set("uid", "x", nx=True) is None
    assert rc.sadd("running_uids", "u1") == 1
    assert rc.sadd("running_uids", "u1") == 0
    assert rc.sadd("running_uids", "u2", "u3") == 2
    assert rc.smembers("running_uids") == {"u1", "u2", "u3"}
    assert rc.srem("running_uids", "u1") == 1
    assert rc.sismember("running_uids", "u1") is False
    assert rc.scard("running_uids") == 2
    assert rc.delete("uid") == 1
    assert rc.exists("uid") == 0


def test_hash_counters(rc):
    assert rc.hsetnx("ps_key", "u", 2) == 1
    assert rc.hsetnx("ps_key", "u", 5) == 0      # idempotent against replayed ADDED
    assert rc.hget("ps_key", "u") == "2"
    assert rc.hincrby("ps_key", "u", -1) == 1
    assert rc.hincrby("ps_key", "u", -1) == 0
    assert rc.hdel("ps_key", "u") == 1
    assert rc.hget("ps_key", "u") is None
    assert rc.hgetall("ps_key") == {}
    rc.hset("h", "a", 1)
    rc.hset("h", "b", "x")
    assert rc.hgetall("h") == {"a": "1", "b": "x"}


def test_list_queue_and_blpop(rc):
    assert rc.rpush("q", "a", "b") == 2
    assert rc.llen("q") == 2
    assert rc.lrange("q", 0, -1) == ["a", "b"]
    assert rc.blpop("q", 1) == ("q", "a")
    assert rc.lpop("q") == "b"
    t0 = time.time()
    assert rc.blpop("q", 0.3) is None
    assert 0.25 < time.time() - t0 < 2.0


def test_blpop_wakes_on_push():
    with StoreServer("127.0.0.1", 0) as srv:
        consumer, producer = StoreClient(srv.host, srv.port), StoreClient(srv.host, srv.port)
        got = []
        t = threading.Thread(target=lambda: got.append(consumer.blpop("dq", 0)))
        t.start()
        time.sleep(0.2)
        assert not got
        producer.rpush("dq", "job")
        t.join(3)
        assert got == [("dq", "job")]


def test_wrong_type_is_an_error_not_a_crash(rc):
    rc.set("k", "v")
    with pytest.raises(Exception):
        rc.sadd("k", "x")
    assert rc.get("k") == "v"


def test_pipeline(rc):
    with rc.pipeline() as p:
        p.sadd("running_uids", "u9")
        p.set("u9", "{}")
        res = p.execute()
    assert len(res) == 2
    assert rc.get("u9") == "{}"


def test_hincrby_is_atomic_across_clients():
    with StoreServer("127.0.0.1", 0) as srv:
        def work():
            c = StoreClient(srv.host, srv.port)
            for _ in range(200):
                c.hincrby("h", "n", 1)
        ts = [threading.Thread(target=work) for _ in range(4)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        assert StoreClient(srv.host, srv.port).hget("h", "n") == "800"


def test_native_server_blpop_wakeup_and_atomic_counters():
    """dtf-stored (C++ epoll server): BLPOP 0 wakes on RPUSH, counters stay atomic under 4 clients."""
    from distributed_tensorflow_amd.native.build import spawn_store

    proc, port = spawn_store()
    try:
        consumer, producer = StoreClient("127.0.0.1", port), StoreClient("127.0.0.1", port)
        got = []
        t = threading.Thread(target=lambda: got.append(consumer.blpop(["a", "dq"], 0)))
        t.start()
        time.sleep(0.2)
        assert not got
        producer.rpush("dq", "job1", "job2")
        t.join(3)
        assert got == [("dq", "job1")]
        assert producer.lrange("dq", 0, -1) == ["job2"]

        def work():
            c = StoreClient("127.0.0.1", port)
            for _ in range(500):
                c.hincrby("h", "n", 1)
        ts = [threading.Thread(target=work) for _ in range(4)]
        [x.start() for x in ts]
        [x.join() for x in ts]
        assert producer.hget("h", "n") == "2000"
        with producer.pipeline() as p:
            p.sadd("s", "x")
            p.set("k", "v")
            assert len(p.execute()) == 2
    finally:
        proc.kill()
        proc.wait()
