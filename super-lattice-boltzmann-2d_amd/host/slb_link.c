/*
 * slb_link.c -- point-to-point messages between the processes of a slab run (one process
 * per GPU).  No counterpart in the reference, which runs on one GPU
 * (reference: src/boltzmann_solver.c:77 selects that GPU).
 *
 * Two transports behind one small interface:
 *   RCCL  -- ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd on a HIP stream:
 *            device buffers travel GPU to GPU over xGMI, nothing blocks the host.  The
 *            library is opened at run time (librccl.so.1), so that a single-GPU program
 *            never loads it.
 *   fds   -- a socket per peer; messages are staged through host memory and the calling
 *            thread blocks until they have arrived.  For rehearsals with several ranks on
 *            one GPU, which RCCL refuses.
 * Messages are matched by order: both ends of a pair issue their sends and receives for
 * each other in the same order.
 */
#include <dlfcn.h>
#include <errno.h>
#include <hip/hip_runtime_api.h>
#include <poll.h>
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <unistd.h>

#include "slb_host.h"

#define HIP_OK(call) HandleError((int)(call), __FILE__, __LINE__)

/* ---- RCCL entry points, resolved at run time ------------------------------------------ */
static struct {
  void *handle;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *);
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
  ncclResult_t (*CommCount)(const ncclComm_t, int *);
  ncclResult_t (*CommDestroy)(ncclComm_t);
  ncclResult_t (*GroupStart)(void);
  ncclResult_t (*GroupEnd)(void);
  ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
  const char *(*GetErrorString)(ncclResult_t);
} rccl;

static int rccl_open(void)
{
  if (rccl.handle) return 0;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (int i = 0; i < 3 && !rccl.handle; i++) rccl.handle = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!rccl.handle) { fprintf(stderr, "ERROR: cannot open librccl.so.1: %s\n", dlerror()); return -1; }
#define SYM(field, name) \
  do { *(void **)&rccl.field = dlsym(rccl.handle, name); \
       if (!rccl.field) { fprintf(stderr, "ERROR: librccl has no %s\n", name); return -1; } } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommCount, "ncclCommCount");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(GroupStart, "ncclGroupStart");
  SYM(GroupEnd, "ncclGroupEnd");
  SYM(Send, "ncclSend");
  SYM(Recv, "ncclRecv");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  return 0;
}

static void rccl_ok(ncclResult_t rc, int line)
{
  if (rc != ncclSuccess) {
    fprintf(stderr, "ERROR: RCCL: %s (%s line %d)\n", rccl.GetErrorString(rc), __FILE__, line);
    exit(EXIT_FAILURE);
  }
}
#define RCCL_OK(call) rccl_ok((call), __LINE__)

/* ---- the link ------------------------------------------------------------------------- */
#define LINK_MAX_OPS 16
struct slb_link {
  int rank, nranks, is_rccl;
  ncclComm_t comm;
  int *fds;                                 /* fds transport: one socket per peer, -1 for self */
  struct { int send, peer; float *dev; size_t n; float *host; } op[LINK_MAX_OPS];
  int n_ops, in_group;
  float *stage;                             /* fds transport: host staging of a group's messages */
  size_t stage_floats;
};

int slb_link_unique_id(char unique_id[128])
{
  if (rccl_open() != 0) return -1;
  ncclUniqueId id;
  if (sizeof(id) != 128) { fprintf(stderr, "ERROR: ncclUniqueId is %zu bytes, not 128\n", sizeof(id)); return -1; }
  RCCL_OK(rccl.GetUniqueId(&id));
  memcpy(unique_id, &id, 128);
  return 0;
}

slb_link *slb_link_open_rccl(int rank, int nranks, const char unique_id[128])
{
  if (rccl_open() != 0) return NULL;
  slb_link *l = (slb_link *)calloc(1, sizeof(*l));
  if (!l) return NULL;
  l->rank = rank; l->nranks = nranks; l->is_rccl = 1;
  ncclUniqueId id;
  memcpy(&id, unique_id, 128);
  RCCL_OK(rccl.CommInitRank(&l->comm, nranks, id, rank));     /* the current HIP device is this rank's */
  return l;
}

slb_link *slb_link_open_fds(int rank, int nranks, const int *fds)
{
  slb_link *l = (slb_link *)calloc(1, sizeof(*l));
  if (!l) return NULL;
  l->rank = rank; l->nranks = nranks; l->is_rccl = 0;
  l->fds = (int *)malloc(sizeof(int) * (size_t)nranks);
  if (!l->fds) { free(l); return NULL; }
  for (int i = 0; i < nranks; i++) l->fds[i] = fds[i];
  return l;
}

int slb_link_rank(const slb_link *l)   { return l->rank; }
int slb_link_nranks(const slb_link *l) { return l->nranks; }
const char *slb_link_transport(const slb_link *l) { return l->is_rccl ? "rccl" : "fds"; }

/* how many ranks the transport itself reports */
int slb_link_ranks_seen(const slb_link *l)
{
  if (!l->is_rccl) return l->nranks;
  int n = 0;
  RCCL_OK(rccl.CommCount(l->comm, &n));
  return n;
}

void slb_link_begin(slb_link *l)
{
  l->n_ops = 0; l->in_group = 1;
  if (l->is_rccl) RCCL_OK(rccl.GroupStart());
}

static void link_op(slb_link *l, int send, int peer, float *dev, size_t n, hipStream_t st)
{
  /* (RCCL matches a rank's message to itself inside a group; a socket to oneself does not exist) */
  if (!l->in_group || peer < 0 || peer >= l->nranks || (peer == l->rank && !l->is_rccl) || l->n_ops >= LINK_MAX_OPS) {
    fprintf(stderr, "ERROR: slb_link: bad message (peer %d of %d, %d in this group)\n", peer, l->nranks, l->n_ops);
    exit(EXIT_FAILURE);
  }
  if (n == 0) return;
  if (l->is_rccl) {
    if (send) RCCL_OK(rccl.Send(dev, n, ncclFloat, peer, l->comm, st));
    else      RCCL_OK(rccl.Recv(dev, n, ncclFloat, peer, l->comm, st));
    return;
  }
  l->op[l->n_ops].send = send; l->op[l->n_ops].peer = peer; l->op[l->n_ops].dev = dev; l->op[l->n_ops].n = n;
  l->op[l->n_ops].host = NULL;
  l->n_ops++;
}

/* st: the stream the group will be closed on (slb_link_end gets the same one) */
void slb_link_send(slb_link *l, int peer, const float *dev, size_t n, void *st)
{
  link_op(l, 1, peer, (float *)dev, n, (hipStream_t)st);
}
void slb_link_recv(slb_link *l, int peer, float *dev, size_t n, void *st)
{
  link_op(l, 0, peer, dev, n, (hipStream_t)st);
}

static void peer_gone(int err)
{
  fprintf(stderr, "ERROR: slb_link: a peer went away (%s)\n", err ? strerror(err) : "end of stream");
  exit(EXIT_FAILURE);
}

/* All messages of a group over the sockets.  Sends and receives make progress side by side
 * (poll + non-blocking transfers), so two neighbours whose messages exceed the socket buffers
 * cannot block each other; of the messages of one pair in one direction only the oldest
 * unfinished one moves, because messages are matched by order. */
static void fds_transfer(slb_link *l)
{
  size_t done[LINK_MAX_OPS];
  int left = 0;
  for (int i = 0; i < l->n_ops; i++) { done[i] = 0; left++; }
  while (left > 0) {
    struct pollfd pf[LINK_MAX_OPS];
    int op_of[LINK_MAX_OPS], np = 0;
    for (int i = 0; i < l->n_ops; i++) {
      if (done[i] == sizeof(float) * l->op[i].n) continue;
      int earlier = 0;
      for (int k = 0; k < i && !earlier; k++)
        earlier = done[k] < sizeof(float) * l->op[k].n && l->op[k].peer == l->op[i].peer && l->op[k].send == l->op[i].send;
      if (earlier) continue;
      pf[np].fd = l->fds[l->op[i].peer];
      pf[np].events = l->op[i].send ? POLLOUT : POLLIN;
      pf[np].revents = 0;
      op_of[np++] = i;
    }
    if (poll(pf, (nfds_t)np, -1) < 0) {
      if (errno == EINTR) continue;
      peer_gone(errno);
    }
    for (int k = 0; k < np; k++) {
      if (!pf[k].revents) continue;
      const int i = op_of[k];
      char *p = (char *)l->op[i].host + done[i];
      const size_t rest = sizeof(float) * l->op[i].n - done[i];
      const ssize_t got = l->op[i].send ? send(pf[k].fd, p, rest, MSG_DONTWAIT | MSG_NOSIGNAL)
                                        : recv(pf[k].fd, p, rest, MSG_DONTWAIT);
      if (got < 0 && (errno == EAGAIN || errno == EWOULDBLOCK || errno == EINTR)) continue;
      if (got <= 0) peer_gone(got < 0 ? errno : 0);
      done[i] += (size_t)got;
      if (done[i] == sizeof(float) * l->op[i].n) left--;
    }
  }
}

void slb_link_end(slb_link *l, void *stream)
{
  hipStream_t st = (hipStream_t)stream;
  l->in_group = 0;
  if (l->is_rccl) { RCCL_OK(rccl.GroupEnd()); return; }
  /* host-staged: what the stream has produced so far goes out, what comes in is on the
   * device before this returns */
  size_t total = 0;
  for (int i = 0; i < l->n_ops; i++) total += l->op[i].n;
  if (total > l->stage_floats) {
    free(l->stage);
    l->stage = (float *)malloc(sizeof(float) * total);
    if (!l->stage) { fprintf(stderr, "ERROR: slb_link: out of host memory for %zu floats\n", total); exit(EXIT_FAILURE); }
    l->stage_floats = total;
  }
  total = 0;
  for (int i = 0; i < l->n_ops; i++) {
    l->op[i].host = l->stage + total;
    total += l->op[i].n;
    if (l->op[i].send)
      HIP_OK(hipMemcpyAsync(l->op[i].host, l->op[i].dev, sizeof(float) * l->op[i].n, hipMemcpyDeviceToHost, st));
  }
  HIP_OK(hipStreamSynchronize(st));
  fds_transfer(l);
  for (int i = 0; i < l->n_ops; i++)
    if (!l->op[i].send)
      HIP_OK(hipMemcpyAsync(l->op[i].dev, l->op[i].host, sizeof(float) * l->op[i].n, hipMemcpyHostToDevice, st));
  HIP_OK(hipStreamSynchronize(st));
  l->n_ops = 0;
}

void slb_link_close(slb_link *l)
{
  if (!l) return;
  if (l->is_rccl && l->comm) rccl.CommDestroy(l->comm);
  if (l->fds) {
    for (int i = 0; i < l->nranks; i++) if (l->fds[i] >= 0) close(l->fds[i]);
    free(l->fds);
  }
  free(l->stage);
  free(l);
}
