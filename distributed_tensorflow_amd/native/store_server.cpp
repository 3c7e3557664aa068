// dtf-stored: native job-store server speaking RESP2 (the Redis wire protocol) for the command
// subset the control plane uses (external X8: the reference talks to a real Redis
// at [redis] host:port; neither redis nor redis-server exists on the target boxes).
//
// Single-threaded epoll loop (all commands are atomic by construction, like Redis), per-key
// BLPOP waiter queues served in arrival order, no persistence.  Wire-compatible with
// utils/store.py's StoreClient and with redis-py / redis-cli for: PING SET(NX) SETNX GET INCR
// INCRBY DEL EXISTS KEYS FLUSHALL SADD SREM SMEMBERS SISMEMBER SCARD HSET HSETNX HGET HGETALL
// HINCRBY HDEL HEXISTS HLEN RPUSH LPUSH LPOP LLEN LRANGE BLPOP QUIT.
//
//   g++ -O2 -std=c++17 -o dtf-stored store_server.cpp     ;   ./dtf-stored [--host H] [--port P]
#include <arpa/inet.h>
#include <errno.h>
#include <fcntl.h>
#include <fnmatch.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <signal.h>
#include <sys/epoll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <variant>
#include <vector>

using Str = std::string;
using Set = std::set<Str>;
using Hash = std::map<Str, Str>;
using List = std::deque<Str>;
using Value = std::variant<Str, Set, Hash, List>;

static std::unordered_map<Str, Value> db;

struct Conn {
    int fd;
    Str in, out;
    bool blocked = false, closing = false;
    std::vector<Str> wait_keys;
    double deadline = 0;     // 0 = forever
};
static std::unordered_map<int, Conn> conns;
static std::unordered_map<Str, std::deque<int>> waiters;   // key -> fds blocked in BLPOP
static int epfd = -1;

static double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

static Str upper(Str s) { for (auto& c : s) c = (char)toupper((unsigned char)c); return s; }
static void r_ok(Str& o) { o += "+OK\r\n"; }
static void r_int(Str& o, long long v) { o += ":" + std::to_string(v) + "\r\n"; }
static void r_nil(Str& o) { o += "$-1\r\n"; }
static void r_bulk(Str& o, const Str& s) { o += "$" + std::to_string(s.size()) + "\r\n" + s + "\r\n"; }
static void r_err(Str& o, const Str& m) { o += "-" + m + "\r\n"; }
static void r_arr(Str& o, size_t n) { o += "*" + std::to_string(n) + "\r\n"; }
static const char* WRONG = "WRONGTYPE Operation against a key holding the wrong kind of value";

template <class T> static T* typed(const Str& k, bool create, bool& wrong) {
    wrong = false;
    auto it = db.find(k);
    if (it == db.end()) {
        if (!create) return nullptr;
        it = db.emplace(k, T{}).first;
    }
    T* p = std::get_if<T>(&it->second);
    if (!p) wrong = true;
    return p;
}
template <class T> static void gc(const Str& k) {
    auto it = db.find(k);
    if (it != db.end()) if (auto* p = std::get_if<T>(&it->second)) if (p->empty()) db.erase(it);
}

static bool to_ll(const Str& s, long long& v) {
    char* e = nullptr; errno = 0;
    v = strtoll(s.c_str(), &e, 10);
    return errno == 0 && e && *e == 0 && !s.empty();
}

static void flush_out(Conn& c);
static void process(Conn& c);

// try to serve a blocked BLPOP client from any of its keys
static bool serve_blocked(Conn& c) {
    for (auto& k : c.wait_keys) {
        bool wrong;
        List* l = typed<List>(k, false, wrong);
        if (l && !l->empty()) {
            r_arr(c.out, 2); r_bulk(c.out, k); r_bulk(c.out, l->front());
            l->pop_front(); gc<List>(k);
            for (auto& kk : c.wait_keys) {
                auto& q = waiters[kk];
                q.erase(std::remove(q.begin(), q.end(), c.fd), q.end());
                if (q.empty()) waiters.erase(kk);
            }
            c.blocked = false; c.wait_keys.clear();
            return true;
        }
    }
    return false;
}

static void wake(const Str& key) {
    auto it = waiters.find(key);
    if (it == waiters.end()) return;
    std::deque<int> q = it->second;     // copy: serve_blocked edits the queues
    for (int fd : q) {
        auto ci = conns.find(fd);
        if (ci == conns.end()) continue;
        bool wrong;
        List* l = typed<List>(key, false, wrong);
        if (!l || l->empty()) break;
        if (serve_blocked(ci->second)) { process(ci->second); flush_out(ci->second); }
    }
}

static void exec(Conn& c, std::vector<Str>& a) {
    Str& o = c.out;
    if (a.empty()) return;
    const Str cmd = upper(a[0]);
    const size_t n = a.size();
    bool wrong = false;
    auto need = [&](size_t k) { if (n < k) { r_err(o, "ERR wrong number of arguments for '" + cmd + "'"); return false; } return true; };
    if (cmd == "PING") { if (n > 1) r_bulk(o, a[1]); else o += "+PONG\r\n"; }
    else if (cmd == "QUIT") { r_ok(o); c.closing = true; }
    else if (cmd == "SET") {
        if (!need(3)) return;
        bool nx = false;
        for (size_t i = 3; i < n; ++i) if (upper(a[i]) == "NX") nx = true;
        if (nx && db.count(a[1])) r_nil(o); else { db[a[1]] = a[2]; r_ok(o); }
    } else if (cmd == "SETNX") {
        if (!need(3)) return;
        if (db.count(a[1])) r_int(o, 0); else { db[a[1]] = a[2]; r_int(o, 1); }
    } else if (cmd == "GET") {
        if (!need(2)) return;
        Str* s = typed<Str>(a[1], false, wrong);
        if (wrong) r_err(o, WRONG); else if (!s) r_nil(o); else r_bulk(o, *s);
    } else if (cmd == "INCR" || cmd == "INCRBY") {
        if (!need(cmd == "INCR" ? 2 : 3)) return;
        long long d = 1, v = 0;
        if (cmd == "INCRBY" && !to_ll(a[2], d)) { r_err(o, "ERR value is not an integer or out of range"); return; }
        Str* s = typed<Str>(a[1], true, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        if (!s->empty() && !to_ll(*s, v)) { r_err(o, "ERR value is not an integer or out of range"); return; }
        *s = std::to_string(v + d); r_int(o, v + d);
    } else if (cmd == "DEL") { long long k = 0; for (size_t i = 1; i < n; ++i) k += db.erase(a[i]); r_int(o, k); }
    else if (cmd == "EXISTS") { long long k = 0; for (size_t i = 1; i < n; ++i) k += db.count(a[i]); r_int(o, k); }
    else if (cmd == "KEYS") {
        if (!need(2)) return;
        std::vector<Str> ks;
        for (auto& kv : db) if (fnmatch(a[1].c_str(), kv.first.c_str(), 0) == 0) ks.push_back(kv.first);
        std::sort(ks.begin(), ks.end());
        r_arr(o, ks.size()); for (auto& k : ks) r_bulk(o, k);
    } else if (cmd == "FLUSHALL" || cmd == "FLUSHDB") { db.clear(); r_ok(o); }
    else if (cmd == "SADD" || cmd == "SREM") {
        if (!need(3)) return;
        Set* s = typed<Set>(a[1], cmd == "SADD", wrong);
        if (wrong) { r_err(o, WRONG); return; }
        long long k = 0;
        if (s) for (size_t i = 2; i < n; ++i) k += (cmd == "SADD") ? (long long)s->insert(a[i]).second : (long long)s->erase(a[i]);
        gc<Set>(a[1]); r_int(o, k);
    } else if (cmd == "SMEMBERS") {
        if (!need(2)) return;
        Set* s = typed<Set>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_arr(o, s ? s->size() : 0); if (s) for (auto& m : *s) r_bulk(o, m);
    } else if (cmd == "SISMEMBER") {
        if (!need(3)) return;
        Set* s = typed<Set>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_int(o, s && s->count(a[2]) ? 1 : 0);
    } else if (cmd == "SCARD") {
        if (!need(2)) return;
        Set* s = typed<Set>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_int(o, s ? (long long)s->size() : 0);
    } else if (cmd == "HSET") {
        if (!need(4) || (n % 2) != 0) { if (n >= 4) r_err(o, "ERR wrong number of arguments for 'HSET'"); return; }
        Hash* h = typed<Hash>(a[1], true, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        long long k = 0;
        for (size_t i = 2; i + 1 < n; i += 2) { k += h->count(a[i]) ? 0 : 1; (*h)[a[i]] = a[i + 1]; }
        r_int(o, k);
    } else if (cmd == "HSETNX") {
        if (!need(4)) return;
        Hash* h = typed<Hash>(a[1], true, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        if (h->count(a[2])) r_int(o, 0); else { (*h)[a[2]] = a[3]; r_int(o, 1); }
    } else if (cmd == "HGET") {
        if (!need(3)) return;
        Hash* h = typed<Hash>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        if (!h || !h->count(a[2])) r_nil(o); else r_bulk(o, (*h)[a[2]]);
    } else if (cmd == "HGETALL") {
        if (!need(2)) return;
        Hash* h = typed<Hash>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_arr(o, h ? 2 * h->size() : 0);
        if (h) for (auto& kv : *h) { r_bulk(o, kv.first); r_bulk(o, kv.second); }
    } else if (cmd == "HINCRBY") {
        if (!need(4)) return;
        long long d, v = 0;
        if (!to_ll(a[3], d)) { r_err(o, "ERR value is not an integer or out of range"); return; }
        Hash* h = typed<Hash>(a[1], true, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        auto it = h->find(a[2]);
        if (it != h->end() && !to_ll(it->second, v)) { r_err(o, "ERR hash value is not an integer"); return; }
        (*h)[a[2]] = std::to_string(v + d); r_int(o, v + d);
    } else if (cmd == "HDEL") {
        if (!need(3)) return;
        Hash* h = typed<Hash>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        long long k = 0;
        if (h) for (size_t i = 2; i < n; ++i) k += h->erase(a[i]);
        gc<Hash>(a[1]); r_int(o, k);
    } else if (cmd == "HEXISTS") {
        if (!need(3)) return;
        Hash* h = typed<Hash>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_int(o, h && h->count(a[2]) ? 1 : 0);
    } else if (cmd == "HLEN") {
        if (!need(2)) return;
        Hash* h = typed<Hash>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_int(o, h ? (long long)h->size() : 0);
    } else if (cmd == "RPUSH" || cmd == "LPUSH") {
        if (!need(3)) return;
        List* l = typed<List>(a[1], true, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        for (size_t i = 2; i < n; ++i) { if (cmd == "RPUSH") l->push_back(a[i]); else l->push_front(a[i]); }
        r_int(o, (long long)l->size());
        wake(a[1]);
    } else if (cmd == "LPOP") {
        if (!need(2)) return;
        List* l = typed<List>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        if (!l || l->empty()) r_nil(o); else { r_bulk(o, l->front()); l->pop_front(); gc<List>(a[1]); }
    } else if (cmd == "LLEN") {
        if (!need(2)) return;
        List* l = typed<List>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        r_int(o, l ? (long long)l->size() : 0);
    } else if (cmd == "LRANGE") {
        if (!need(4)) return;
        long long s, e;
        if (!to_ll(a[2], s) || !to_ll(a[3], e)) { r_err(o, "ERR value is not an integer or out of range"); return; }
        List* l = typed<List>(a[1], false, wrong);
        if (wrong) { r_err(o, WRONG); return; }
        long long len = l ? (long long)l->size() : 0;
        if (s < 0) s = std::max(0LL, len + s);
        if (e < 0) e = len + e;
        e = std::min(e, len - 1);
        if (!l || s > e) { r_arr(o, 0); return; }
        r_arr(o, (size_t)(e - s + 1));
        for (long long i = s; i <= e; ++i) r_bulk(o, (*l)[(size_t)i]);
    } else if (cmd == "BLPOP") {
        if (!need(3)) return;
        double to = atof(a[n - 1].c_str());
        c.wait_keys.assign(a.begin() + 1, a.end() - 1);
        for (auto& k : c.wait_keys) { typed<List>(k, false, wrong); if (wrong) { r_err(o, WRONG); c.wait_keys.clear(); return; } }
        c.blocked = true;
        if (serve_blocked(c)) return;
        c.deadline = to > 0 ? now_s() + to : 0;
        for (auto& k : c.wait_keys) waiters[k].push_back(c.fd);
    } else {
        r_err(o, "ERR unknown command '" + a[0] + "'");
    }
}

// parse as many complete commands as the input buffer holds; stops while the client is blocked
static void process(Conn& c) {
    while (!c.blocked && !c.closing) {
        Str& in = c.in;
        if (in.empty()) return;
        std::vector<Str> args;
        size_t pos = 0;
        if (in[0] == '*') {
            size_t e = in.find("\r\n");
            if (e == Str::npos) return;
            long long cnt = atoll(in.c_str() + 1);
            pos = e + 2;
            bool complete = true;
            for (long long i = 0; i < cnt; ++i) {
                if (pos >= in.size() || in[pos] != '$') { complete = pos < in.size() ? (c.closing = true, false) : false; break; }
                size_t e2 = in.find("\r\n", pos);
                if (e2 == Str::npos) { complete = false; break; }
                long long len = atoll(in.c_str() + pos + 1);
                size_t start = e2 + 2;
                if (in.size() < start + (size_t)len + 2) { complete = false; break; }
                args.emplace_back(in, start, (size_t)len);
                pos = start + (size_t)len + 2;
            }
            if (!complete) return;
        } else {  // inline command
            size_t e = in.find('\n');
            if (e == Str::npos) return;
            Str line = in.substr(0, e);
            pos = e + 1;
            size_t i = 0;
            while (i < line.size()) {
                while (i < line.size() && isspace((unsigned char)line[i])) ++i;
                size_t j = i;
                while (j < line.size() && !isspace((unsigned char)line[j])) ++j;
                if (j > i) args.emplace_back(line, i, j - i);
                i = j;
            }
        }
        in.erase(0, pos);
        exec(c, args);
    }
}

static void flush_out(Conn& c) {
    while (!c.out.empty()) {
        ssize_t w = ::send(c.fd, c.out.data(), c.out.size(), MSG_NOSIGNAL);
        if (w > 0) c.out.erase(0, (size_t)w);
        else if (w < 0 && (errno == EAGAIN || errno == EWOULDBLOCK)) {
            epoll_event ev{}; ev.events = EPOLLIN | EPOLLOUT; ev.data.fd = c.fd;
            epoll_ctl(epfd, EPOLL_CTL_MOD, c.fd, &ev);
            return;
        } else { c.closing = true; return; }
    }
    epoll_event ev{}; ev.events = EPOLLIN; ev.data.fd = c.fd;
    epoll_ctl(epfd, EPOLL_CTL_MOD, c.fd, &ev);
}

static void drop(int fd) {
    auto it = conns.find(fd);
    if (it == conns.end()) return;
    for (auto& k : it->second.wait_keys) {
        auto& q = waiters[k];
        q.erase(std::remove(q.begin(), q.end(), fd), q.end());
        if (q.empty()) waiters.erase(k);
    }
    epoll_ctl(epfd, EPOLL_CTL_DEL, fd, nullptr);
    ::close(fd);
    conns.erase(it);
}

int main(int argc, char** argv) {
    const char* host = "127.0.0.1";
    int port = 30009;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--host") && i + 1 < argc) host = argv[++i];
        else if (!strcmp(argv[i], "--port") && i + 1 < argc) port = atoi(argv[++i]);
    }
    signal(SIGPIPE, SIG_IGN);
    int ls = socket(AF_INET, SOCK_STREAM | SOCK_NONBLOCK, 0);
    int one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in sa{};
    sa.sin_family = AF_INET; sa.sin_port = htons((uint16_t)port);
    inet_pton(AF_INET, host, &sa.sin_addr);
    if (bind(ls, (sockaddr*)&sa, sizeof sa) || listen(ls, 256)) { perror("dtf-stored: bind/listen"); return 1; }
    socklen_t sl = sizeof sa;
    getsockname(ls, (sockaddr*)&sa, &sl);
    printf("dtf-stored listening on %s:%d\n", host, ntohs(sa.sin_port));
    fflush(stdout);
    epfd = epoll_create1(0);
    epoll_event ev{}; ev.events = EPOLLIN; ev.data.fd = ls;
    epoll_ctl(epfd, EPOLL_CTL_ADD, ls, &ev);
    std::vector<epoll_event> evs(256);
    char buf[65536];
    for (;;) {
        int nev = epoll_wait(epfd, evs.data(), (int)evs.size(), 100);
        for (int i = 0; i < nev; ++i) {
            int fd = evs[i].data.fd;
            if (fd == ls) {
                for (;;) {
                    int cfd = accept4(ls, nullptr, nullptr, SOCK_NONBLOCK);
                    if (cfd < 0) break;
                    setsockopt(cfd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
                    conns[cfd].fd = cfd;
                    epoll_event cev{}; cev.events = EPOLLIN; cev.data.fd = cfd;
                    epoll_ctl(epfd, EPOLL_CTL_ADD, cfd, &cev);
                }
                continue;
            }
            auto it = conns.find(fd);
            if (it == conns.end()) continue;
            Conn& c = it->second;
            if (evs[i].events & (EPOLLHUP | EPOLLERR)) { drop(fd); continue; }
            if (evs[i].events & EPOLLIN) {
                for (;;) {
                    ssize_t r = ::recv(fd, buf, sizeof buf, 0);
                    if (r > 0) c.in.append(buf, (size_t)r);
                    else if (r == 0) { c.closing = true; break; }
                    else { if (errno != EAGAIN && errno != EWOULDBLOCK) c.closing = true; break; }
                }
                process(c);
            }
            flush_out(c);
            if (c.closing && c.out.empty()) drop(fd);
        }
        // BLPOP timeouts
        double t = now_s();
        std::vector<int> expired;
        for (auto& kv : conns) if (kv.second.blocked && kv.second.deadline > 0 && t >= kv.second.deadline) expired.push_back(kv.first);
        for (int fd : expired) {
            Conn& c = conns[fd];
            for (auto& k : c.wait_keys) {
                auto& q = waiters[k];
                q.erase(std::remove(q.begin(), q.end(), fd), q.end());
                if (q.empty()) waiters.erase(k);
            }
            c.blocked = false; c.wait_keys.clear();
            c.out += "*-1\r\n";
            process(c);         // pipelined commands behind the BLPOP
            flush_out(c);
        }
    }
}
