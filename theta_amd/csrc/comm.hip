// The one communication step of a sharded search, owned by the library (include/theta_hip.h, "several GPUs").
//
// The candidate space shards by rank range with no data-path collective; what replaces the reference's merge of its
// workers' lists (find_mins, RunTHetA.py:107-122, after the Queue fan-out of RunTHetA.py:124-171) is ONE exchange at the
// end: all-reduce(min) of the shard minima, then an all-gather of the finalists within the window of the global minimum
// (a few hundred bytes per GPU -- latency bound, the xGMI bandwidth is irrelevant).
//
// One process per GPU.  Bootstrap is a TCP star the library sets up itself (rank 0 listens on addr:port, the others
// connect): it carries rank 0's ncclUniqueId to everybody, after which every rank joins ncclCommInitRank on its context's
// GPU and the collectives run over RCCL (xGMI between the GPUs of a node) on the context's stream.  librccl.so is loaded
// on first use (dlopen): a single-GPU process never pays for its 570 MB.  Transport HOST keeps the TCP star for the
// collectives too -- for the 2-process tests on machines without GPUs; same entry points, same merge code.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <errno.h>
#include <math.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <vector>

#include <rccl/rccl.h>     // types and prototypes only: the symbols are resolved with dlsym, not linked

#include "common.hpp"

namespace {

struct Rccl {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;            // (optional: rccl_wait)
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
};
Rccl g_rccl;

int rccl_load() {
    if (g_rccl.handle) return THETA_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *nm : names) {
        h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
        if (h) break;
    }
    if (!h) {
        theta_set_error("cannot load librccl.so: %s", dlerror());
        return THETA_ERR_HIP;
    }
#define SYM(field, name)                                            \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);          \
    if (!g_rccl.field) {                                            \
        theta_set_error("librccl.so lacks %s", name);               \
        dlclose(h);                                                 \
        return THETA_ERR_HIP;                                       \
    }
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllReduce, "ncclAllReduce")
    SYM(AllGather, "ncclAllGather")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(GetVersion, "ncclGetVersion")
#undef SYM
    g_rccl.CommAbort = (decltype(g_rccl.CommAbort))dlsym(h, "ncclCommAbort");
    g_rccl.handle = h;
    return THETA_OK;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t _r = (expr);                                                                        \
        if (_r != ncclSuccess) {                                                                         \
            theta_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return THETA_ERR_HIP;                                                                        \
        }                                                                                                \
    } while (0)

int send_all(int fd, const void *buf, size_t n) {
    const char *p = (const char *)buf;
    while (n) {
        ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) {
            theta_set_error("comm: send failed: %s", strerror(errno));
            return THETA_ERR_HIP;
        }
        p += k;
        n -= (size_t)k;
    }
    return THETA_OK;
}

int recv_all(int fd, void *buf, size_t n) {
    char *p = (char *)buf;
    while (n) {
        ssize_t k = ::recv(fd, p, n, 0);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) {
            theta_set_error("comm: receive failed: %s", k == 0 ? "peer closed the connection" : strerror(errno));
            return THETA_ERR_HIP;
        }
        p += k;
        n -= (size_t)k;
    }
    return THETA_OK;
}

// seconds a rank waits for the others at the RENDEZVOUS (THETA_COMM_TIMEOUT_S, default 300): a rank that never shows up must
// not hang the others for ever
int rendezvous_timeout_s() {
    if (const char *e = getenv("THETA_COMM_TIMEOUT_S")) {
        const int v = atoi(e);
        if (v > 0) return v;
    }
    return 300;
}
void set_timeouts(int fd, int seconds) {
    struct timeval tv;
    tv.tv_sec = seconds;    // 0: block for ever
    tv.tv_usec = 0;
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
    setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}
void tune_socket(int fd) {
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    set_timeouts(fd, rendezvous_timeout_s());
}

}   // namespace

struct theta_comm {
    theta_ctx *ctx = nullptr;
    int device = 0;                // (= ctx->device: the destructor must not need the context)
    int rank = 0, world = 1, transport = THETA_COMM_RCCL;
    // TCP star: rank 0 holds one socket per peer (index = peer rank), the others one socket to rank 0
    std::vector<int> peers;
    int up = -1;
    ncclComm_t nccl = nullptr;
    DevBuf d_send, d_recv;
    uint64_t collectives = 0;      // collectives issued (diagnostic)
    bool dead_stream = false;      // a collective was abandoned without ncclCommAbort: never synchronise the device for this communicator again
};

static void close_star(theta_comm *c) {
    for (int fd : c->peers)
        if (fd >= 0) ::close(fd);
    c->peers.clear();
    if (c->up >= 0) ::close(c->up);
    c->up = -1;
}

static int star_connect(theta_comm *c, const char *addr, int port) {
    if (c->world == 1) return THETA_OK;
    struct addrinfo hints, *res = nullptr;
    memset(&hints, 0, sizeof(hints));
    hints.ai_family = AF_INET;
    hints.ai_socktype = SOCK_STREAM;
    char ports[16];
    snprintf(ports, sizeof(ports), "%d", port);
    if (getaddrinfo(addr && *addr ? addr : "127.0.0.1", ports, &hints, &res) != 0 || !res) {
        theta_set_error("comm: cannot resolve %s", addr ? addr : "(null)");
        return THETA_ERR_ARG;
    }
    int rc = THETA_OK;
    if (c->rank == 0) {
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        struct sockaddr_in sa;
        memset(&sa, 0, sizeof(sa));
        sa.sin_family = AF_INET;
        sa.sin_port = htons((uint16_t)port);
        sa.sin_addr.s_addr = htonl(INADDR_ANY);
        if (::bind(ls, (struct sockaddr *)&sa, sizeof(sa)) != 0 || ::listen(ls, c->world) != 0) {
            theta_set_error("comm: rank 0 cannot listen on port %d: %s", port, strerror(errno));
            ::close(ls);
            freeaddrinfo(res);
            return THETA_ERR_HIP;
        }
        const int tmo = rendezvous_timeout_s();
        struct timeval tv;
        tv.tv_sec = tmo;
        tv.tv_usec = 0;
        setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
        c->peers.assign(c->world, -1);
        for (int k = 1; k < c->world && rc == THETA_OK; k++) {
            int fd = ::accept(ls, nullptr, nullptr);
            if (fd < 0) {
                if (errno == EAGAIN || errno == EWOULDBLOCK)
                    theta_set_error("comm: only %d of %d ranks reached rank 0 on port %d within %d s (THETA_COMM_TIMEOUT_S): a rank "
                                    "failed to start, or MASTER_ADDR / the port differ between ranks", k, c->world, port, tmo);
                else
                    theta_set_error("comm: accept failed: %s", strerror(errno));
                rc = THETA_ERR_HIP;
                break;
            }
            tune_socket(fd);
            int32_t hello[2] = {0, 0};
            rc = recv_all(fd, hello, sizeof(hello));
            if (rc == THETA_OK && (hello[0] < 1 || hello[0] >= c->world || hello[1] != c->world || c->peers[hello[0]] >= 0)) {
                theta_set_error("comm: unexpected peer (rank %d of %d)", hello[0], hello[1]);
                rc = THETA_ERR_ARG;
            }
            if (rc == THETA_OK) c->peers[hello[0]] = fd;
            else ::close(fd);
        }
        ::close(ls);
    } else {
        int fd = -1;
        const double t_end = (double)time(nullptr) + (double)rendezvous_timeout_s();
        while (true) {
            fd = ::socket(AF_INET, SOCK_STREAM, 0);
            if (::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
            ::close(fd);
            fd = -1;
            if ((double)time(nullptr) > t_end) break;
            usleep(50 * 1000);      // rank 0 may not be listening yet
        }
        if (fd < 0) {
            theta_set_error("comm: rank %d cannot reach rank 0 at %s:%d within %d s (THETA_COMM_TIMEOUT_S)", c->rank, addr, port,
                            rendezvous_timeout_s());
            rc = THETA_ERR_HIP;
        } else {
            tune_socket(fd);
            int32_t hello[2] = {c->rank, c->world};
            rc = send_all(fd, hello, sizeof(hello));
            c->up = fd;
        }
    }
    freeaddrinfo(res);
    if (rc != THETA_OK) close_star(c);
    return rc;
}

// The rendezvous is over: the sockets now carry the collectives of a search whose shards may reach the exchange many minutes
// apart -- no time-out there (THETA_COMM_COLLECTIVE_TIMEOUT_S sets one, in seconds, for debugging).
static void star_collective_mode(theta_comm *c) {
    int tmo = 0;
    if (const char *e = getenv("THETA_COMM_COLLECTIVE_TIMEOUT_S")) tmo = atoi(e) > 0 ? atoi(e) : 0;
    for (int fd : c->peers)
        if (fd >= 0) set_timeouts(fd, tmo);
    if (c->up >= 0) set_timeouts(c->up, tmo);
}

// rank 0's buffer to everybody (TCP star)
static int star_bcast(theta_comm *c, void *buf, size_t n) {
    if (c->world == 1) return THETA_OK;
    if (c->rank == 0) {
        for (int k = 1; k < c->world; k++) {
            int rc = send_all(c->peers[k], buf, n);
            if (rc) return rc;
        }
        return THETA_OK;
    }
    return recv_all(c->up, buf, n);
}

// everybody's `bytes` to everybody, rank-major in recv (TCP star: gather at rank 0, broadcast)
static int star_allgather(theta_comm *c, const void *send, size_t bytes, void *recv) {
    char *out = (char *)recv;
    if (c->rank == 0) {
        if (bytes) memcpy(out, send, bytes);
        for (int k = 1; k < c->world; k++) {
            int rc = recv_all(c->peers[k], out + (size_t)k * bytes, bytes);
            if (rc) return rc;
        }
    } else {
        int rc = send_all(c->up, send, bytes);
        if (rc) return rc;
    }
    return star_bcast(c, out, bytes * (size_t)c->world);
}

extern "C" int theta_comm_create(theta_ctx *ctx, int rank, int world, const char *addr, int port, int transport,
                                 theta_comm **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || (transport != THETA_COMM_RCCL && transport != THETA_COMM_HOST)) {
        theta_set_error("theta_comm_create: bad argument (rank %d, world %d, transport %d)", rank, world, transport);
        return THETA_ERR_ARG;
    }
    if (transport == THETA_COMM_RCCL && !ctx) {
        theta_set_error("theta_comm_create: the RCCL transport needs a context (a GPU)");
        return THETA_ERR_ARG;
    }
    if (world > 1 && (port < 1 || port > 65535)) {
        theta_set_error("theta_comm_create: bad port %d", port);
        return THETA_ERR_ARG;
    }
    theta_comm *c = new theta_comm();
    c->ctx = ctx;
    c->device = ctx ? ctx->device : 0;
    c->rank = rank;
    c->world = world;
    c->transport = transport;
    int rc = star_connect(c, addr, port);
    if (rc) {
        delete c;
        return rc;
    }
    if (transport == THETA_COMM_RCCL) {
        rc = rccl_load();
        ncclUniqueId id;
        memset(&id, 0, sizeof(id));
        int32_t okflag = rc == THETA_OK;
        if (rc == THETA_OK && rank == 0) {
            ncclResult_t r = g_rccl.GetUniqueId(&id);
            if (r != ncclSuccess) {
                theta_set_error("ncclGetUniqueId failed: %s", g_rccl.GetErrorString(r));
                okflag = 0;
                rc = THETA_ERR_HIP;
            }
        }
        // EVERY rank's status is gathered (a rank whose librccl.so does not load must not let the others wait inside
        // ncclCommInitRank for ever), then rank 0's id travels
        std::vector<int32_t> flags((size_t)world, 0);
        int rc2 = star_allgather(c, &okflag, sizeof(okflag), flags.data());
        int bad_rank = -1;
        for (int k = 0; k < world && rc2 == THETA_OK; k++)
            if (!flags[k] && bad_rank < 0) bad_rank = k;
        if (rc2 == THETA_OK && bad_rank < 0) rc2 = star_bcast(c, &id, sizeof(id));
        if (rc == THETA_OK && rc2 != THETA_OK) rc = rc2;
        if (rc == THETA_OK && bad_rank >= 0) {
            theta_set_error("comm: rank %d could not set up RCCL (librccl.so / ncclGetUniqueId): no rank joins the communicator", bad_rank);
            rc = THETA_ERR_HIP;
        }
        // from here on everything goes over RCCL; the star stays open, silent, as the LIVENESS channel of the collectives: a rank
        // that dies (out of memory, a HIP abort) closes its end, and the ranks waiting for it inside a collective notice (rccl_wait)
        if (rc != THETA_OK) close_star(c);
        if (rc == THETA_OK) {
            hipError_t e = hipSetDevice(ctx->device);
            if (e != hipSuccess) {
                theta_set_error("hipSetDevice failed: %s", hipGetErrorString(e));
                rc = THETA_ERR_HIP;
            }
        }
        if (rc == THETA_OK) {
            ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, id, rank);
            fflush(stdout);                      // (RCCL's version banner goes to the C stdout: out now, not at exit behind the caller's own last line)
            if (r != ncclSuccess) {
                theta_set_error("ncclCommInitRank(rank %d of %d) failed: %s", rank, world, g_rccl.GetErrorString(r));
                rc = THETA_ERR_HIP;
            }
        }
        if (rc) {
            close_star(c);
            delete c;
            return rc;
        }
    }
    if (transport == THETA_COMM_HOST) star_collective_mode(c);
    *out = c;
    return THETA_OK;
}

extern "C" void theta_comm_destroy(theta_comm *c) {
    if (!c) return;
    if (c->nccl) {
        (void)hipSetDevice(c->device);
        (void)hipDeviceSynchronize();
        (void)g_rccl.CommDestroy(c->nccl);
        (void)hipGetLastError();
    }
    close_star(c);
    if (c->dead_stream) {          // (hipFree waits for the device: the staging buffers of an abandoned collective are leaked with it)
        c->d_send.p = c->d_recv.p = nullptr;
        c->d_send.bytes = c->d_recv.bytes = 0;
    }
    delete c;
}

extern "C" int theta_comm_info(theta_comm *c, int *rank, int *world, int *transport, int *rccl_version, uint64_t *collectives) {
    if (!c) {
        theta_set_error("null communicator");
        return THETA_ERR_ARG;
    }
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (transport) *transport = c->transport;
    if (rccl_version) {
        int v = 0;
        if (c->transport == THETA_COMM_RCCL && g_rccl.GetVersion) (void)g_rccl.GetVersion(&v);
        *rccl_version = v;
    }
    if (collectives) *collectives = c->collectives;
    return THETA_OK;
}

// Wait for the stream an RCCL collective was issued on -- with an eye on the other ranks.  hipStreamSynchronize alone waits for ever
// when a rank has died (its peers' kernels spin on a flag nobody will set): the reference's counterpart, the result queue of
// RunTHetA.py:96-105, at least ends with the parent.  Here the rendezvous sockets stay open and silent; while the stream is
// busy they are polled, and a hang-up (the kernel closes a dead process's sockets) starts a grace period -- a rank that has
// FINISHED its last collective may legitimately close first, and then this rank's stream completes within milliseconds --
// after which the communicator is aborted (ncclCommAbort: the kernels leave), the star is closed so that the remaining ranks
// notice in turn, and the call fails.  THETA_COMM_GRACE_S (default 20) is that period; THETA_COMM_COLLECTIVE_TIMEOUT_S, if
// set, bounds the whole wait.  world = 1: a plain synchronisation.
static double now_s() {
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return (double)tv.tv_sec + 1e-6 * (double)tv.tv_usec;
}
static int env_seconds(const char *name, int dflt) {
    if (const char *e = getenv(name)) {
        const int v = atoi(e);
        if (v > 0) return v;
    }
    return dflt;
}
static int rccl_wait(theta_comm *c, hipStream_t st) {
    if (c->world == 1 || (c->peers.empty() && c->up < 0)) {
        HIP_TRY(hipStreamSynchronize(st));
        return THETA_OK;
    }
    const double t0 = now_s();
    const int grace = env_seconds("THETA_COMM_GRACE_S", 20), limit = env_seconds("THETA_COMM_COLLECTIVE_TIMEOUT_S", 0);
    double t_gone = -1.0;
    int gone_rank = -1;
    std::vector<struct pollfd> fds;
    std::vector<int> who;
    if (c->rank == 0) {
        for (int k = 1; k < (int)c->peers.size(); k++)
            if (c->peers[k] >= 0) {
                fds.push_back({c->peers[k], (short)(POLLIN | POLLRDHUP), 0});
                who.push_back(k);
            }
    } else if (c->up >= 0) {
        fds.push_back({c->up, (short)(POLLIN | POLLRDHUP), 0});
        who.push_back(0);
    }
    int spins = 0;
    while (true) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return THETA_OK;
        if (q != hipErrorNotReady) {
            theta_set_error("comm: the collective's stream failed: %s", hipGetErrorString(q));
            (void)hipGetLastError();
            return THETA_ERR_HIP;
        }
        if (++spins < 200) continue;                       // (a collective of a few bytes: done within microseconds)
        if (t_gone < 0.0 && !fds.empty()) {
            const int n = ::poll(fds.data(), (nfds_t)fds.size(), 5);
            for (size_t i = 0; n > 0 && i < fds.size() && t_gone < 0.0; i++) {
                bool gone = (fds[i].revents & (POLLHUP | POLLERR | POLLRDHUP | POLLNVAL)) != 0;
                if (!gone && (fds[i].revents & POLLIN)) {
                    char b;
                    gone = ::recv(fds[i].fd, &b, 1, MSG_PEEK | MSG_DONTWAIT) == 0;
                }
                if (gone) {
                    t_gone = now_s();
                    gone_rank = who[i];
                }
            }
        } else {
            struct timespec ts = {0, 5 * 1000 * 1000};
            nanosleep(&ts, nullptr);
        }
        const double t = now_s();
        const bool expired = limit > 0 && t - t0 > (double)limit;
        if ((t_gone >= 0.0 && t - t_gone > (double)grace) || expired) {
            if (expired)
                theta_set_error("comm: rank %d waited %d s inside a collective (THETA_COMM_COLLECTIVE_TIMEOUT_S): communicator aborted", c->rank, limit);
            else
                theta_set_error("comm: rank %d left the job while rank %d was waiting for it inside a collective (its process ended: out of "
                                "memory? a HIP abort?): communicator aborted", gone_rank, c->rank);
            close_star(c);                                 // (the remaining ranks notice in turn)
            if (c->nccl && g_rccl.CommAbort) {
                (void)g_rccl.CommAbort(c->nccl);
                c->nccl = nullptr;
                (void)hipStreamSynchronize(st);
            } else if (c->nccl) {
                // a librccl without ncclCommAbort: the stuck collective cannot be taken off the stream, and ncclCommDestroy on
                // such a communicator may never return.  The communicator is LEAKED (marked dead: later collectives fail, the
                // destructor skips it, nothing waits on the stream) -- the process is expected to exit on this error.
                c->nccl = nullptr;
                c->dead_stream = true;
            }
            (void)hipGetLastError();
            return THETA_ERR_HIP;
        }
    }
}

static int stage(theta_comm *c, size_t send_bytes, size_t recv_bytes) {
    int rc;
    if (c->d_send.bytes < send_bytes && (rc = c->d_send.alloc(std::max<size_t>(send_bytes, 4096)))) return rc;
    if (c->d_recv.bytes < recv_bytes && (rc = c->d_recv.alloc(std::max<size_t>(recv_bytes, 4096)))) return rc;
    return THETA_OK;
}

extern "C" int theta_comm_allgather(theta_comm *c, const void *send, size_t bytes, void *recv) {
    if (!c || !recv || (bytes && !send)) {
        theta_set_error("theta_comm_allgather: null argument");
        return THETA_ERR_ARG;
    }
    c->collectives++;
    if (bytes == 0) return THETA_OK;
    if (c->transport == THETA_COMM_HOST) return star_allgather(c, send, bytes, recv);
    if (!c->nccl) {
        theta_set_error("comm: the communicator was aborted (a rank left the job)");
        return THETA_ERR_HIP;
    }
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    int rc = stage(c, bytes, bytes * (size_t)c->world);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_send.p, send, bytes, hipMemcpyHostToDevice, st));
    NCCL_TRY(g_rccl.AllGather(c->d_send.p, c->d_recv.p, bytes, ncclUint8, c->nccl, st));
    HIP_TRY(hipMemcpyAsync(recv, c->d_recv.p, bytes * (size_t)c->world, hipMemcpyDeviceToHost, st));
    return rccl_wait(c, st);
}

// op: 0 = min, 1 = sum, 2 = max; in place on host doubles
static int allreduce(theta_comm *c, double *v, int count, int op) {
    if (!c || (count > 0 && !v) || count < 0) {
        theta_set_error("theta_comm_allreduce: bad argument");
        return THETA_ERR_ARG;
    }
    c->collectives++;
    if (count == 0) return THETA_OK;
    if (c->transport == THETA_COMM_HOST) {
        std::vector<double> all((size_t)count * c->world);
        int rc = star_allgather(c, v, (size_t)count * sizeof(double), all.data());
        if (rc) return rc;
        for (int i = 0; i < count; i++) {
            double acc = all[i];
            for (int k = 1; k < c->world; k++) {
                const double x = all[(size_t)k * count + i];
                acc = op == 0 ? fmin(acc, x) : (op == 1 ? acc + x : fmax(acc, x));
            }
            v[i] = acc;
        }
        return THETA_OK;
    }
    if (!c->nccl) {
        theta_set_error("comm: the communicator was aborted (a rank left the job)");
        return THETA_ERR_HIP;
    }
    HIP_TRY(hipSetDevice(c->ctx->device));
    hipStream_t st = c->ctx->stream;
    const size_t bytes = (size_t)count * sizeof(double);
    int rc = stage(c, bytes, bytes);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_send.p, v, bytes, hipMemcpyHostToDevice, st));
    NCCL_TRY(g_rccl.AllReduce(c->d_send.p, c->d_recv.p, (size_t)count, ncclDouble, op == 0 ? ncclMin : (op == 1 ? ncclSum : ncclMax),
                              c->nccl, st));
    HIP_TRY(hipMemcpyAsync(v, c->d_recv.p, bytes, hipMemcpyDeviceToHost, st));
    return rccl_wait(c, st);
}

extern "C" int theta_comm_allreduce_min(theta_comm *c, double *v, int count) { return allreduce(c, v, count, 0); }
extern "C" int theta_comm_allreduce_sum(theta_comm *c, double *v, int count) { return allreduce(c, v, count, 1); }
extern "C" int theta_comm_allreduce_max(theta_comm *c, double *v, int count) { return allreduce(c, v, count, 2); }

extern "C" int theta_comm_barrier(theta_comm *c) {
    double z = 0.0;
    return allreduce(c, &z, 1, 1);
}

// ---- the exchange itself ----------------------------------------------------------------------------------------
// Replaces find_mins (RunTHetA.py:107-122).  Every rank passes the finalists of its shard (reference-order values from
// theta_solve_batch); every rank gets back, in rank order, the finalists of ALL shards that lie within `window` of the
// global minimum -- plus every record whose NLL is NaN: the reference's isClose treats NaN as close (Misc.py:44-46), so
// such a record interacts with the running minimum wherever it stands, and the host's replay needs them all.
extern "C" int theta_exchange_finalists(theta_comm *c, int n, int m, int count, const double *nll, const double *mu,
                                        const uint64_t *rank, const uint8_t *C, const double *vals, double window, int cap,
                                        double *o_nll, double *o_mu, uint64_t *o_rank, uint8_t *o_C, double *o_vals,
                                        int *n_out, double *global_min) {
    if (!c || !n_out || count < 0 || (n != 2 && n != 3) || m < 1 || (count > 0 && (!nll || !mu || !rank || !C || !vals)) ||
        !(window >= 0.0)) {
        theta_set_error("theta_exchange_finalists: bad argument");
        return THETA_ERR_ARG;
    }
    double gmin = INFINITY;
    for (int i = 0; i < count; i++)
        if (nll[i] == nll[i]) gmin = fmin(gmin, nll[i]);
    // (the smallest capacity any rank offers rides along: running out of room has to be a COLLECTIVE decision, or the ranks
    // that still fit would go on to the all-gather alone)
    double red[2] = {gmin, (double)cap};
    int rc = theta_comm_allreduce_min(c, red, 2);
    if (rc) return rc;
    gmin = red[0];
    const size_t min_cap = red[1] > 0.0 ? (size_t)red[1] : 0;
    if (global_min) *global_min = gmin;
    std::vector<int> keep;
    for (int i = 0; i < count; i++)
        if (!(nll[i] == nll[i]) || nll[i] <= gmin + window) keep.push_back(i);
    std::vector<double> counts(c->world, 0.0);
    {
        // (a double carries a count exactly; one collective type fewer)
        std::vector<double> mine(c->world, 0.0);
        mine[c->rank] = (double)keep.size();
        rc = theta_comm_allreduce_sum(c, mine.data(), c->world);
        if (rc) return rc;
        counts = mine;
    }
    size_t total = 0, widest = 0;
    for (int k = 0; k < c->world; k++) {
        total += (size_t)counts[k];
        widest = std::max(widest, (size_t)counts[k]);
    }
    *n_out = (int)total;
    if (total == 0) return THETA_OK;
    if (min_cap < total || !o_nll || !o_mu || !o_rank || !o_C || !o_vals) {
        theta_set_error("%zu finalists after the exchange but capacity is %d", total, cap);
        return THETA_ERR_CAPACITY;
    }
    // fixed-size records: nll, mu[n], vals[m], rank[2] as 8-byte words, then the matrix bytes padded to 8
    const size_t cb = (size_t)m * (n - 1), words = 1 + (size_t)n + (size_t)m + 2 + (cb + 7) / 8, rb = words * 8;
    std::vector<unsigned char> sendbuf(widest * rb, 0), recvbuf(widest * rb * (size_t)c->world, 0);
    for (size_t k = 0; k < keep.size(); k++) {
        const int i = keep[k];
        unsigned char *p = sendbuf.data() + k * rb;
        memcpy(p, &nll[i], 8);
        memcpy(p + 8, &mu[(size_t)i * n], 8 * (size_t)n);
        memcpy(p + 8 + 8 * (size_t)n, &vals[(size_t)i * m], 8 * (size_t)m);
        memcpy(p + 8 + 8 * (size_t)(n + m), &rank[2 * (size_t)i], 16);
        memcpy(p + 8 + 8 * (size_t)(n + m) + 16, &C[(size_t)i * cb], cb);
    }
    rc = theta_comm_allgather(c, sendbuf.data(), widest * rb, recvbuf.data());
    if (rc) return rc;
    struct Ref {
        uint64_t lo, hi;
        const unsigned char *p;
    };
    std::vector<Ref> refs;
    for (int k = 0; k < c->world; k++)
        for (size_t j = 0; j < (size_t)counts[k]; j++) {
            const unsigned char *p = recvbuf.data() + ((size_t)k * widest + j) * rb;
            Ref r;
            memcpy(&r.lo, p + 8 + 8 * (size_t)(n + m), 8);
            memcpy(&r.hi, p + 8 + 8 * (size_t)(n + m) + 8, 8);
            r.p = p;
            refs.push_back(r);
        }
    std::stable_sort(refs.begin(), refs.end(), [](const Ref &a, const Ref &b) { return a.hi != b.hi ? a.hi < b.hi : a.lo < b.lo; });
    for (size_t k = 0; k < refs.size(); k++) {
        const unsigned char *p = refs[k].p;
        memcpy(&o_nll[k], p, 8);
        memcpy(&o_mu[k * n], p + 8, 8 * (size_t)n);
        memcpy(&o_vals[k * m], p + 8 + 8 * (size_t)n, 8 * (size_t)m);
        o_rank[2 * k] = refs[k].lo;
        o_rank[2 * k + 1] = refs[k].hi;
        memcpy(&o_C[k * cb], p + 8 + 8 * (size_t)(n + m) + 16, cb);
    }
    return THETA_OK;
}
