/* TEST INFRASTRUCTURE — a stand-in librccl.so for a box without GPUs.  Never loaded by the product unless COMET_RCCL_LIBRARY names it.
 *
 * It exports the eleven entry points csrc/exchange_rccl.hpp binds (ncclGetUniqueId, ncclCommInitRank, ncclCommDestroy, ncclCommCount,
 * ncclCommUserRank, ncclSend, ncclRecv, ncclAllGather, ncclGroupStart, ncclGroupEnd, ncclGetErrorString) with the signatures of nccl.h 2.x and
 * moves the bytes between PROCESSES over TCP on 127.0.0.1 — "device" pointers are host pointers here.  What it pins about the caller
 * (tests/test_rccl_shim_procs_cpu.py runs the product's RcclTransportT against it in 2 and 8 processes):
 *   - argument order and the ncclDataType_t values (0 int8, 1 uint8, 2 int32, 3 uint32, 4 int64, 5 uint64, 6 half, 7 float, 8 double,
 *     9 bfloat16): a count is turned into bytes with the type's size, and every message carries (source, byte count, sequence number) —
 *     a receive whose posted size differs from what its peer sent is ncclInvalidUsage, as a mismatched pair hangs or corrupts on RCCL;
 *   - group semantics: sends / receives between ncclGroupStart and the matching ncclGroupEnd are queued and progress TOGETHER at the end
 *     (every send on its own thread), so a rank may post all its sends before any receive — and outside a group a send blocks until its
 *     peer receives, which is what deadlocks a caller that forgets the group;
 *   - a rank sending to itself inside a group;  ncclAllGather's sendcount in ELEMENTS per rank;  ncclCommCount / ncclCommUserRank.
 * The unique id holds the TCP port rank 0's rendezvous listens on (the id is created on rank 0 and travels out of band, like RCCL's). */
#define _GNU_SOURCE
#include <arpa/inet.h>
#include <errno.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#define MAXR 64
enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
typedef struct { char internal[128]; } ncclUniqueId;

struct ncclComm {
  int world, rank;
  int fd[MAXR];                 /* fd[p]: the connection to rank p (−1 for myself) */
  uint64_t seq_out[MAXR], seq_in[MAXR];
  /* log of what the caller did, for the test to read back through fake_rccl_counters */
  int64_t n_allgather, n_groups, n_send, n_recv, n_self, bytes_out, bytes_in;
};
typedef struct ncclComm* ncclComm_t;

static __thread char t_err[256];
static int g_listen_fd = -1;           /* rank 0's rendezvous socket, bound when the id is made */
static int g_listen_port = 0;

static int fail(int rc, const char* msg) {
  snprintf(t_err, sizeof t_err, "%s (errno %d: %s)", msg, errno, strerror(errno));
  fprintf(stderr, "fake_rccl: %s\n", t_err);
  return rc;
}

static int type_size(int dt) {
  switch (dt) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return -1;
  }
}

static int write_all(int fd, const void* p, size_t n) {
  const char* c = (const char*)p;
  while (n) {
    ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k < 0) { if (errno == EINTR) continue; return -1; }
    c += k; n -= (size_t)k;
  }
  return 0;
}
static int read_all(int fd, void* p, size_t n) {
  char* c = (char*)p;
  while (n) {
    struct pollfd pf = {fd, POLLIN, 0};
    int pr = poll(&pf, 1, 60000);
    if (pr == 0) { errno = ETIMEDOUT; return -1; }
    if (pr < 0) { if (errno == EINTR) continue; return -1; }
    ssize_t k = recv(fd, c, n, 0);
    if (k == 0) { errno = ECONNRESET; return -1; }
    if (k < 0) { if (errno == EINTR) continue; return -1; }
    c += k; n -= (size_t)k;
  }
  return 0;
}

static int listen_any(int* port) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  if (fd < 0) return -1;
  int one = 1;
  setsockopt(fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  struct sockaddr_in a;
  memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  a.sin_port = 0;
  if (bind(fd, (struct sockaddr*)&a, sizeof a) < 0 || listen(fd, MAXR) < 0) { close(fd); return -1; }
  socklen_t l = sizeof a;
  getsockname(fd, (struct sockaddr*)&a, &l);
  *port = ntohs(a.sin_port);
  return fd;
}
static int connect_to(int port) {
  for (int attempt = 0; attempt < 600; attempt++) {
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    if (fd < 0) return -1;
    struct sockaddr_in a;
    memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    a.sin_port = htons((uint16_t)port);
    if (connect(fd, (struct sockaddr*)&a, sizeof a) == 0) {
      int one = 1;
      setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      return fd;
    }
    close(fd);
    usleep(50000);
  }
  return -1;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) {
    case ncclSuccess: return "no error";
    case ncclSystemError: return t_err[0] ? t_err : "unhandled system error";
    case ncclInvalidArgument: return t_err[0] ? t_err : "invalid argument";
    case ncclInvalidUsage: return t_err[0] ? t_err : "invalid usage";
    default: return t_err[0] ? t_err : "internal error";
  }
}

int ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return fail(ncclInvalidArgument, "ncclGetUniqueId: null id");
  memset(id, 0, sizeof *id);
  if (g_listen_fd < 0 && (g_listen_fd = listen_any(&g_listen_port)) < 0) return fail(ncclSystemError, "ncclGetUniqueId: cannot listen");
  snprintf(id->internal, sizeof id->internal, "FAKE-RCCL:%d", g_listen_port);
  return ncclSuccess;
}

/* rendezvous: every rank listens on its own port; ranks 1.. report (rank, port) to rank 0, which answers with the table; then rank j
 * connects to every rank i < j and says who it is */
int ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAXR || rank < 0 || rank >= nranks) return fail(ncclInvalidArgument, "ncclCommInitRank: bad communicator / nranks / rank");
  int port0 = 0;
  if (sscanf(id.internal, "FAKE-RCCL:%d", &port0) != 1) return fail(ncclInvalidArgument, "ncclCommInitRank: not an id of this library");
  struct ncclComm* c = (struct ncclComm*)calloc(1, sizeof *c);
  c->world = nranks; c->rank = rank;
  for (int p = 0; p < MAXR; p++) c->fd[p] = -1;
  if (nranks == 1) { *out = c; return ncclSuccess; }
  int myport = 0, lfd = -1;
  int32_t ports[MAXR];
  memset(ports, 0, sizeof ports);
  if (rank == 0) {
    if (g_listen_fd < 0 || g_listen_port != port0) return fail(ncclInvalidUsage, "ncclCommInitRank: rank 0 must be the process that made the id");
    lfd = g_listen_fd; myport = port0;
    ports[0] = port0;
    int ctl[MAXR];
    for (int k = 1; k < nranks; k++) {
      int fd = accept(lfd, NULL, NULL);
      int32_t hello[2];
      if (fd < 0 || read_all(fd, hello, sizeof hello) < 0 || hello[0] < 1 || hello[0] >= nranks) return fail(ncclSystemError, "ncclCommInitRank: rendezvous accept");
      ports[hello[0]] = hello[1];
      ctl[hello[0]] = fd;
    }
    for (int k = 1; k < nranks; k++) { if (write_all(ctl[k], ports, sizeof ports) < 0) return fail(ncclSystemError, "ncclCommInitRank: rendezvous answer"); close(ctl[k]); }
  } else {
    if ((lfd = listen_any(&myport)) < 0) return fail(ncclSystemError, "ncclCommInitRank: cannot listen");
    int fd = connect_to(port0);
    int32_t hello[2] = {rank, myport};
    if (fd < 0 || write_all(fd, hello, sizeof hello) < 0 || read_all(fd, ports, sizeof ports) < 0) return fail(ncclSystemError, "ncclCommInitRank: rendezvous with rank 0");
    close(fd);
  }
  for (int i = 0; i < rank; i++) {           /* I connect to every lower rank … */
    int fd = connect_to(ports[i]);
    int32_t me = rank;
    if (fd < 0 || write_all(fd, &me, 4) < 0) return fail(ncclSystemError, "ncclCommInitRank: connect to a peer");
    c->fd[i] = fd;
  }
  for (int k = rank + 1; k < nranks; k++) {  /* … and accept every higher one */
    int fd = accept(lfd, NULL, NULL);
    int32_t who = -1;
    if (fd < 0 || read_all(fd, &who, 4) < 0 || who <= rank || who >= nranks || c->fd[who] >= 0) return fail(ncclSystemError, "ncclCommInitRank: accept a peer");
    int one = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
    c->fd[who] = fd;
  }
  if (rank != 0) close(lfd);
  *out = c;
  return ncclSuccess;
}

int ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  for (int p = 0; p < MAXR; p++) if (c->fd[p] >= 0) close(c->fd[p]);
  free(c);
  return ncclSuccess;
}
int ncclCommCount(const ncclComm_t c, int* n) { if (!c || !n) return fail(ncclInvalidArgument, "ncclCommCount"); *n = c->world; return ncclSuccess; }
int ncclCommUserRank(const ncclComm_t c, int* r) { if (!c || !r) return fail(ncclInvalidArgument, "ncclCommUserRank"); *r = c->rank; return ncclSuccess; }

/* ---- point to point, queued inside a group ---- */
struct op { int is_send; void* buf; size_t bytes; int peer; struct ncclComm* comm; int rc; };
static __thread int t_depth = 0;
static __thread struct op* t_ops = NULL;
static __thread int t_nops = 0, t_cap = 0;

struct hdr { uint32_t magic; int32_t src; uint64_t bytes, seq; };
static struct op* t_ops_shared;      /* the group being run, for its send threads (one group runs at a time per process in these tests) */
static int t_nops_shared;
static __thread int t_internal = 0;  /* inside ncclAllGather: its own point-to-point traffic is not the caller's */

/* one thread per PEER: that peer's sends of the group, in posting order (two threads must never interleave on one connection) */
struct peer_job { int peer; int rc; pthread_t th; int started; };
static void* send_main(void* arg) {
  struct peer_job* j = (struct peer_job*)arg;
  struct op* ops = t_ops_shared;
  for (int i = 0; i < t_nops_shared; i++) {
    struct op* o = &ops[i];
    if (!o->is_send || o->rc == -1 || o->peer != j->peer) continue;
    struct ncclComm* c = o->comm;
    struct hdr h = {0x52434346u, c->rank, o->bytes, c->seq_out[o->peer]++};
    if (write_all(c->fd[o->peer], &h, sizeof h) < 0 || write_all(c->fd[o->peer], o->buf, o->bytes) < 0) { j->rc = ncclSystemError; break; }
  }
  return NULL;
}
static int recv_now(struct op* o) {
  struct ncclComm* c = o->comm;
  struct hdr h;
  if (read_all(c->fd[o->peer], &h, sizeof h) < 0) return fail(ncclSystemError, "ncclRecv: the peer went away or stayed silent");
  if (h.magic != 0x52434346u || h.src != o->peer || h.seq != c->seq_in[o->peer]) return fail(ncclInternalError, "ncclRecv: stream out of step");
  c->seq_in[o->peer]++;
  if (h.bytes != o->bytes) {
    snprintf(t_err, sizeof t_err, "ncclRecv from rank %d: %llu bytes posted, the peer sent %llu (mismatched send / recv pair)", o->peer, (unsigned long long)o->bytes, (unsigned long long)h.bytes);
    fprintf(stderr, "fake_rccl: %s\n", t_err);
    return ncclInvalidUsage;
  }
  if (read_all(c->fd[o->peer], o->buf, o->bytes) < 0) return fail(ncclSystemError, "ncclRecv: short message");
  return ncclSuccess;
}

static int run_ops(void) {
  int rc = ncclSuccess;
  /* a rank's message to itself: pair the sends and receives in posting order */
  for (int i = 0; i < t_nops; i++) {
    struct op* s = &t_ops[i];
    if (!s->is_send || s->peer != s->comm->rank) continue;
    int found = 0;
    for (int j = 0; j < t_nops && !found; j++) {
      struct op* r = &t_ops[j];
      if (r->is_send || r->peer != r->comm->rank || r->comm != s->comm || r->rc == -1) continue;
      if (r->bytes != s->bytes) return fail(ncclInvalidUsage, "self send / recv sizes differ");
      memmove(r->buf, s->buf, s->bytes);
      r->rc = -1;                /* consumed */
      if (!t_internal) s->comm->n_self++;
      found = 1;
    }
    if (!found) return fail(ncclInvalidUsage, "a send to myself without a matching recv in the same group");
    s->rc = -1;
  }
  for (int i = 0; i < t_nops; i++) if (!t_ops[i].is_send && t_ops[i].peer == t_ops[i].comm->rank && t_ops[i].rc != -1) return fail(ncclInvalidUsage, "a recv from myself without a matching send in the same group");
  struct peer_job jobs[MAXR];
  memset(jobs, 0, sizeof jobs);
  t_ops_shared = t_ops;
  t_nops_shared = t_nops;
  for (int i = 0; i < t_nops; i++) {
    struct op* o = &t_ops[i];
    if (o->rc == -1 || !o->is_send || jobs[o->peer].started) continue;
    jobs[o->peer].peer = o->peer;
    jobs[o->peer].started = 1;
    if (pthread_create(&jobs[o->peer].th, NULL, send_main, &jobs[o->peer]) != 0) return fail(ncclSystemError, "pthread_create");
  }
  for (int i = 0; i < t_nops; i++) {
    struct op* o = &t_ops[i];
    if (o->rc == -1 || o->is_send) continue;
    int r = recv_now(o);
    if (r != ncclSuccess && rc == ncclSuccess) rc = r;
  }
  for (int p = 0; p < MAXR; p++) {
    if (!jobs[p].started) continue;
    pthread_join(jobs[p].th, NULL);
    if (jobs[p].rc > 0 && rc == ncclSuccess) rc = fail(jobs[p].rc, "ncclSend: the peer went away");
  }
  t_nops = 0;
  return rc;
}

static int post(int is_send, void* buf, size_t count, int dt, int peer, ncclComm_t c) {
  const int ts = type_size(dt);
  if (!c) return fail(ncclInvalidArgument, "null communicator");
  if (ts < 0) return fail(ncclInvalidArgument, "not an ncclDataType_t");
  if (peer < 0 || peer >= c->world) return fail(ncclInvalidArgument, "peer outside the communicator");
  if (count && !buf) return fail(ncclInvalidArgument, "null buffer with a non-zero count");
  if (t_nops == t_cap) { t_cap = t_cap ? 2 * t_cap : 32; t_ops = (struct op*)realloc(t_ops, (size_t)t_cap * sizeof *t_ops); }
  struct op o = {is_send, buf, count * (size_t)ts, peer, c, 0};
  t_ops[t_nops++] = o;
  if (!t_internal) {
    if (is_send) { c->n_send++; if (peer != c->rank) c->bytes_out += (int64_t)o.bytes; } else { c->n_recv++; if (peer != c->rank) c->bytes_in += (int64_t)o.bytes; }
  }
  return t_depth ? ncclSuccess : run_ops();      /* outside a group the call completes — blocks — here */
}

int ncclGroupStart(void) { t_depth++; return ncclSuccess; }
int ncclGroupEnd(void) {
  if (t_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
  if (--t_depth) return ncclSuccess;
  if (t_nops && !t_internal) t_ops[0].comm->n_groups++;
  return run_ops();
}
int ncclSend(const void* buf, size_t count, int dt, int peer, ncclComm_t c, void* stream) { (void)stream; return post(1, (void*)buf, count, dt, peer, c); }
int ncclRecv(void* buf, size_t count, int dt, int peer, ncclComm_t c, void* stream) { (void)stream; return post(0, buf, count, dt, peer, c); }

/* sendcount ELEMENTS from every rank, rank r's at recvbuf + r · sendcount · size */
int ncclAllGather(const void* sendbuf, void* recvbuf, size_t sendcount, int dt, ncclComm_t c, void* stream) {
  (void)stream;
  const int ts = type_size(dt);
  if (!c || ts < 0 || (sendcount && (!sendbuf || !recvbuf))) return fail(ncclInvalidArgument, "ncclAllGather: bad argument");
  if (t_depth) return fail(ncclInvalidUsage, "ncclAllGather inside a group is not something this stand-in models");
  const size_t bytes = sendcount * (size_t)ts;
  c->n_allgather++;
  t_internal = 1;
  int rc = ncclGroupStart();
  for (int p = 0; p < c->world && rc == ncclSuccess; p++) {
    rc = ncclSend(sendbuf, sendcount, dt, p, c, stream);
    if (rc == ncclSuccess) rc = ncclRecv((char*)recvbuf + (size_t)p * bytes, sendcount, dt, p, c, stream);
  }
  int rc2 = ncclGroupEnd();
  t_internal = 0;
  return rc != ncclSuccess ? rc : rc2;
}

/* what the caller did so far: {allgathers, groups, sends, recvs, self pairs, bytes out, bytes in} */
void fake_rccl_counters(ncclComm_t c, int64_t* out7) {
  out7[0] = c->n_allgather; out7[1] = c->n_groups; out7[2] = c->n_send; out7[3] = c->n_recv; out7[4] = c->n_self; out7[5] = c->bytes_out; out7[6] = c->bytes_in;
}
