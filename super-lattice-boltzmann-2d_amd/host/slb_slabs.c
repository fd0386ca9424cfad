/*
 * slb_slabs.c -- the time loop over phi_y slabs, one process per GPU, in C: launch order,
 * halo columns and the chain of running sums of av(), with the messages going through
 * slb_link (RCCL send/recv on a HIP stream, or sockets for rehearsals on one GPU).
 * The reference runs on one GPU; the loop this spreads over several is
 * reference src/boltzmann_solver.c:196-302, its display=4 record :348-380.
 *
 * Per launch of S = 2 or 3 loop iterations (slb_step_fused2, slb_step_multi) and slab:
 *
 *   side stream:  edge column ranges -> pack -> { send/recv 11 columns per neighbour,
 *                 + totals of av() on their way down the chain } -> unpack
 *   main stream:  interior column range
 *
 * Each stream waits for the other's PREVIOUS launch only (results go to arrays the present
 * lattices are not in), so interior launches follow each other without a gap and the few-KB
 * messages travel meanwhile.  Lattices are bit-identical to a single-GPU run.
 *
 * av() over slabs: the reference's three sums are ONE float running sum over all columns
 * (reference: src/boltzmann_c_solver.c:413-437), so slab r starts from the totals slab r-1
 * ended with.  The three rows the sums read are handed out by the double-step launches; the
 * iterations of AVC_BATCH double steps form a batch whose sums run in one launch on a third
 * stream.  A batch's totals leave rank r for rank r+1 in the message group of the
 * AVC_LAG*(r+1)-th exchange after the batch was complete -- a fixed schedule every rank
 * derives by itself, so the few floats ride along with the halo messages, nothing is polled
 * and no rank ever waits on the host.  The last rank applies the running-average updates in
 * step order.
 */
#include <hip/hip_runtime_api.h>
#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "slb_host.h"

#define HIP_OK(call) HandleError((int)(call), __FILE__, __LINE__)

#define AVC_BATCH 4                 /* launches of the time loop per launch of the sums at most     */
#define AVC_LAG   4                 /* exchanges between a batch's arrival on a rank and its leaving */
#define AVC_SETS  8                 /* lattices per batch at most                                  */

typedef struct {
  float *rows[AVC_SETS];            /* [3][pitch] each: a rows 0, 1 and b row 1                    */
  float  t[AVC_SETS];               /* loop times of the lattices                                  */
  int    n;                         /* lattices gathered                                           */
  int    steps;                     /* double steps (or single steps) gathered                     */
  long   ready;                     /* exchange count when the batch was complete; -1: still open  */
  float *init, *tot;                /* device, 3 floats per lattice                                */
  hipEvent_t done;                  /* sums (and on the last rank the updates) finished            */
  int    summed, sent;
} av_batch;

struct slb_slabs {
  slb_solver *s;
  slb_link   *link;
  int rank, nranks, left, right;    /* left, right: the neighbours' ranks in the link, -1: none  */
  hipStream_t main, side, av;
  hipEvent_t  ev_main, ev_side, ev_x;
  int   side_busy;
  float *send2[2], *recv2[2];       /* a launch's message per side: up to 2*11*(N+1) floats         */
  float *send4[2], *recv4[2];       /* G columns of one lattice: 2*G*(N+1) floats                  */
  long  exchanges;                  /* message groups so far (the clock of the av schedule)        */
  /* av chain */
  av_batch *ring;                   /* ring of batches                                             */
  int   ring_n;
  long  batch_open;                 /* index of the batch being filled                             */
  long  batch_next_sum;             /* first batch whose sums have not been launched               */
  long  batch_next_send;            /* first batch whose totals have not left                      */
  float *av_data;                   /* device, 6 floats (meaningful on the last rank)              */
  /* stall watchdog (SLB_WATCHDOG=<seconds>): what the loop was doing when it last moved */
  volatile long ticks;              /* grows with every call into the driver and every exchange    */
  const char *volatile where;       /* the call or wait the host thread entered last               */
  volatile long double_steps, single_steps;
  pthread_t dog;
  int   dog_seconds;
  volatile int dog_stop;
};

#define MARK(r, text) do { (r)->where = (text); (r)->ticks++; } while (0)

static void check(int rc, const char *what)
{
  if (rc > 0) HandleError(rc, __FILE__, __LINE__);
  if (rc < 0) { fprintf(stderr, "ERROR: %s was refused\n", what); exit(EXIT_FAILURE); }
}

static av_batch *batch_at(slb_slabs *r, long b) { return &r->ring[b % r->ring_n]; }

static slb_slabs *slabs_create(int rank, int nranks, slb_link *link, int loopback);

slb_slabs *slb_slabs_create(int rank, int nranks, slb_link *link)
{
  return slabs_create(rank, nranks, link, 0);
}

/* For measuring what the driver and the transport cost on ONE GPU: the slab of rank `rank`
 * of `nranks` with both neighbours' messages addressed to this process itself (link: an
 * RCCL communicator of one rank).  The ghost columns then hold this slab's own edge columns,
 * so the lattice is not the physical one; launches, messages and their ordering are those
 * of a real run.  No av chain. */
slb_slabs *slb_slabs_create_loopback(int rank, int nranks, slb_link *link)
{
  return slabs_create(rank, nranks, link, 1);
}

/* ---- stall watchdog -----------------------------------------------------------------------
 * SLB_WATCHDOG=<seconds>: a thread that looks at the driver's tick count once a second; when
 * it has not moved for that long the rank writes down what it was doing -- the call the host
 * thread is in, the exchange and batch counters of the av schedule, which streams still have
 * work and which of the events that order them have fired -- to stderr (and to
 * $SLB_WATCHDOG_DIR/slab_stall_rank<q>.txt) and ends the process with status 86, so that a
 * stalled run leaves its state behind instead of being killed from outside without a trace. */
static const char *busy_or_idle(hipError_t e)
{
  return e == hipSuccess ? "done" : (e == hipErrorNotReady ? "PENDING" : hipGetErrorString(e));
}

static void dump_state(slb_slabs *r, FILE *f, long idle_s)
{
  fprintf(f, "slab stall: rank %d of %d made no progress for %ld s\n", r->rank, r->nranks, idle_s);
  fprintf(f, "  host thread last entered: %s (tick %ld)\n", r->where ? r->where : "(nothing yet)", r->ticks);
  fprintf(f, "  double steps %ld, single steps %ld, exchanges %ld, side_busy %d, transport %s\n", r->double_steps,
          r->single_steps, r->exchanges, r->side_busy, r->link ? slb_link_transport(r->link) : "none");
  fprintf(f, "  streams: main %s, side %s, av %s\n", busy_or_idle(hipStreamQuery(r->main)),
          busy_or_idle(hipStreamQuery(r->side)), busy_or_idle(hipStreamQuery(r->av)));
  fprintf(f, "  events: ev_main (side waits for it) %s, ev_side (main waits for it) %s, ev_x (av waits for it) %s\n",
          busy_or_idle(hipEventQuery(r->ev_main)), busy_or_idle(hipEventQuery(r->ev_side)),
          busy_or_idle(hipEventQuery(r->ev_x)));
  fprintf(f, "  av chain: batch_open %ld, next_sum %ld, next_send %ld, ring of %d\n", r->batch_open, r->batch_next_sum,
          r->batch_next_send, r->ring_n);
  for (int k = 0; k < r->ring_n; k++) {
    const av_batch *b = &r->ring[k];
    if (!b->init) continue;
    fprintf(f, "    slot %d: lattices %d, steps %d, ready at exchange %ld, summed %d, sent %d, done-event %s\n", k, b->n,
            b->steps, b->ready, b->summed, b->sent, busy_or_idle(hipEventQuery(b->done)));
  }
  fflush(f);
}

static void *watchdog(void *arg)
{
  slb_slabs *r = (slb_slabs *)arg;
  long seen = r->ticks, idle = 0;
  while (!r->dog_stop) {
    struct timespec one = {1, 0};
    nanosleep(&one, NULL);
    if (r->ticks != seen) { seen = r->ticks; idle = 0; continue; }
    if (++idle < r->dog_seconds) continue;
    dump_state(r, stderr, idle);
    const char *dir = getenv("SLB_WATCHDOG_DIR");
    if (dir) {
      char name[512];
      snprintf(name, sizeof(name), "%s/slab_stall_rank%d.txt", dir, r->rank);
      FILE *f = fopen(name, "w");
      if (f) { dump_state(r, f, idle); fclose(f); }
    }
    _exit(86);
  }
  return NULL;
}

static slb_slabs *slabs_create(int rank, int nranks, slb_link *link, int loopback)
{
  if (nranks > 1 && !link) { fprintf(stderr, "slb_slabs_create: %d ranks need a link\n", nranks); return NULL; }
  slb_slabs *r = (slb_slabs *)calloc(1, sizeof(*r));
  if (!r) return NULL;
  r->rank = rank; r->nranks = nranks; r->link = link;
  r->left = rank > 0 ? rank - 1 : -1;
  r->right = rank < nranks - 1 ? rank + 1 : -1;
  if (loopback) {
    if (r->left >= 0) r->left = slb_link_rank(link);
    if (r->right >= 0) r->right = slb_link_rank(link);
  }
  r->s = slb_solver_create(rank, nranks);
  if (!r->s) { free(r); return NULL; }
  /* not the legacy null stream: its implicit synchronisation with other streams would
   * serialise the halo messages with the interior launches */
  HIP_OK(hipStreamCreateWithFlags(&r->main, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&r->side, hipStreamNonBlocking));
  HIP_OK(hipStreamCreateWithFlags(&r->av, hipStreamNonBlocking));
  HIP_OK(hipEventCreateWithFlags(&r->ev_main, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&r->ev_side, hipEventDisableTiming));
  HIP_OK(hipEventCreateWithFlags(&r->ev_x, hipEventDisableTiming));
  const size_t col = sizeof(float) * 2 * (size_t)(r->s->N + 1);
  for (int side = 0; side < 2; side++) {
    HIP_OK(hipMalloc((void **)&r->send2[side], 11 * col));
    HIP_OK(hipMalloc((void **)&r->recv2[side], 11 * col));
    HIP_OK(hipMalloc((void **)&r->send4[side], 6 * col));
    HIP_OK(hipMalloc((void **)&r->recv4[side], 6 * col));
  }
  HIP_OK(hipMalloc((void **)&r->av_data, 6 * sizeof(float)));
  HIP_OK(hipMemset(r->av_data, 0, 6 * sizeof(float)));
  /* batches alive at once: one being filled, AVC_LAG exchanges per rank upstream on their
   * way here, one being summed, one waiting to leave */
  r->ring_n = AVC_LAG * nranks / 2 + 4;                 /* (a batch is complete after two launches of three steps) */
  r->ring = (av_batch *)calloc((size_t)r->ring_n, sizeof(av_batch));
  for (int k = 0; k < r->ring_n; k++) {
    r->ring[k].ready = -1;
    HIP_OK(hipEventCreateWithFlags(&r->ring[k].done, hipEventDisableTiming));
  }
  HIP_OK(hipDeviceSynchronize());
  slb_solver_set_stream(r->s, r->main);
  const char *dog = getenv("SLB_WATCHDOG");
  r->dog_seconds = dog ? atoi(dog) : 0;
  if (r->dog_seconds > 0 && pthread_create(&r->dog, NULL, watchdog, r) != 0) r->dog_seconds = 0;
  return r;
}

slb_solver *slb_slabs_solver(slb_slabs *r) { return r->s; }
void *slb_slabs_main_stream(slb_slabs *r)  { return (void *)r->main; }
float *slb_slabs_av_data(slb_slabs *r)     { return r->av_data; }

/* ---- message groups ---------------------------------------------------------------------
 * One group per exchange: the halo buffers given (NULL: none on that side), and whatever
 * the av schedule has for this exchange.  Runs on the side stream. */
static void av_launch_ready(slb_slabs *r);

static void exchange(slb_slabs *r, float *const send[2], float *const recv[2], size_t n)
{
  const long e = r->exchanges;
  av_batch *in = NULL, *out = NULL;
  MARK(r, "exchange: message group");
  /* a batch arrives here AVC_LAG*rank exchanges after it was complete, and leaves AVC_LAG later */
  /* (batches complete in order and `ready` grows from batch to batch, so only the oldest
   * batch not yet summed, or not yet sent, can be due) */
  if (r->left >= 0 && r->batch_next_sum < r->batch_open) {
    av_batch *b = batch_at(r, r->batch_next_sum);
    if (b->ready + (long)AVC_LAG * r->rank == e) in = b;
  }
  if (r->right >= 0 && r->batch_next_send < r->batch_open) {
    av_batch *b = batch_at(r, r->batch_next_send);
    if (b->ready + (long)AVC_LAG * (r->rank + 1) == e) out = b;
  }
  if (r->nranks > 1) {
    if (out) {
      if (!out->summed) { fprintf(stderr, "ERROR: av chain: batch due to leave before its sums were started\n"); exit(EXIT_FAILURE); }
      HIP_OK(hipStreamWaitEvent(r->side, out->done, 0));
    }
    slb_link_begin(r->link);
    if (r->left >= 0 && send) {
      slb_link_send(r->link, r->left, send[0], n, r->side);
      slb_link_recv(r->link, r->left, recv[0], n, r->side);
    }
    if (r->right >= 0 && send) {
      slb_link_send(r->link, r->right, send[1], n, r->side);
      slb_link_recv(r->link, r->right, recv[1], n, r->side);
    }
    if (in)  slb_link_recv(r->link, r->left, in->init, 3 * (size_t)in->n, r->side);
    if (out) slb_link_send(r->link, r->right, out->tot, 3 * (size_t)out->n, r->side);
    slb_link_end(r->link, r->side);
    if (out) { out->sent = 1; r->batch_next_send++; }
  }
  r->exchanges = e + 1;
  if (in) {
    HIP_OK(hipEventRecord(r->ev_x, r->side));
    HIP_OK(hipStreamWaitEvent(r->av, r->ev_x, 0));
    av_launch_ready(r);
  }
}

/* ---- stepping --------------------------------------------------------------------------- */
/* the engine's launches from here on go to this stream (state of this slab's solver
 * object, not of the process) */
#define on_stream(st) slb_solver_set_stream(r->s, (void *)(st))

/* first half step, then the ghost columns of the new half-step lattice (the main
 * lattice's are from the initial state) */
void slb_slabs_tiptoe(slb_slabs *r)
{
  MARK(r, "slb_slabs_tiptoe");
  on_stream(r->side);
  slb_solver_tiptoe(r->s);
  for (int side = 0; side < 2; side++)
    if ((side == 0 ? r->left : r->right) >= 0) check(slb_solver_halo_pack(r->s, 0, side, r->s->G, r->send4[side]), "halo pack");
  exchange(r, r->send4, r->recv4, 2 * (size_t)r->s->G * (size_t)(r->s->N + 1));
  for (int side = 0; side < 2; side++)
    if ((side == 0 ? r->left : r->right) >= 0) check(slb_solver_halo_unpack(r->s, 0, side, r->s->G, r->recv4[side]), "halo unpack");
  HIP_OK(hipStreamSynchronize(r->side));
  on_stream(r->main);
}

/* the launch for one column range: two steps are the double-step kernel's, three the ring kernel's */
static void range_launch(slb_solver *s, int S, int j_first, int j_last, const float *c, float *const *rows)
{
  if (S == 2) slb_solver_double_step_range_rows(s, j_first, j_last, c, rows ? rows[0] : NULL, rows ? rows[1] : NULL);
  else        slb_solver_multi_step_range_rows(s, S, j_first, j_last, c, rows);
}

/* S = 2 or 3 loop iterations in one launch per column range.  rows (NULL or S pointers): [3][pitch]
 * device buffers that receive the rows av() reads of the main lattice after each of the S
 * iterations, at this slab's columns. */
void slb_slabs_multi_step(slb_slabs *r, int S, float *const *rows)
{
  slb_solver *s = r->s;
  if (S != 2 && S != 3) { fprintf(stderr, "ERROR: slb_slabs_multi_step takes 2 or 3 steps per launch\n"); exit(EXIT_FAILURE); }
  const int even = S == 2;
  float c[16];
  MARK(r, S == 2 ? "slb_slabs_multi_step (2)" : "slb_slabs_multi_step (3)");
  r->double_steps++;
  slb_steps_cosines(s->t, S, c);
  /* a neighbour gets this slab's outermost G-1 main and G half-step columns whatever S is (the
   * launch that follows may be one of three steps, which reads that deep): the edge ranges are
   * the G outermost owned columns of the last sweep, the interior the rest */
  const int first = s->G, last_owned = s->G + s->W - 1, E = s->G;
  const int lo = r->left >= 0 ? first + E : first;
  const int hi = r->right >= 0 ? last_owned - E : last_owned;
  if (r->nranks == 1) {
    on_stream(r->main);
    range_launch(s, S, lo, hi, c, rows);
    HIP_OK(hipEventRecord(r->ev_main, r->main));
    exchange(r, NULL, NULL, 0);                       /* the clock of the av schedule ticks all the same */
    if (S == 2) slb_solver_double_step_finish(s); else slb_solver_multi_step_finish(s, S);
    return;
  }
  on_stream(r->main);
  /* two steps write the spare arrays: if stale (first use, or an odd number of steps since) they are
   * copied on the main stream, after whatever the side stream still has to put into the lattices
   * they are copies of.  Three steps write a/b[next], a/b[next_hs], which hold no lattice in use. */
  const int stale = even && (!s->spare_a || s->spare_of != s->current || s->spare_of_hs != s->current_hs);
  if (r->side_busy && stale) HIP_OK(hipStreamWaitEvent(r->main, r->ev_side, 0));
  if (even) slb_solver_prepare_double_step(s);
  /* halo chain: reads what the last interior launch wrote, writes the output arrays' edge and
   * ghost columns (which that launch read, when they were its inputs).  When the spares were
   * just re-copied on the main stream the event has to stand behind those copies: the side
   * stream's edge launches and halo unpack write the same arrays. */
  if (!r->side_busy || stale) HIP_OK(hipEventRecord(r->ev_main, r->main));
  HIP_OK(hipStreamWaitEvent(r->side, r->ev_main, 0));
  on_stream(r->side);
  if (r->left >= 0)  range_launch(s, S, first, first + E - 1, c, rows);
  if (r->right >= 0) range_launch(s, S, last_owned - (E - 1), last_owned, c, rows);
  check(slb_solver_halo_pack_multi(s, S, r->left >= 0 ? r->send2[0] : NULL, r->right >= 0 ? r->send2[1] : NULL), "halo pack");
  exchange(r, r->send2, r->recv2, 2 * (size_t)(2 * E - 1) * (size_t)(s->N + 1));
  /* interior: reads what the last halo chain wrote, writes the output arrays' interior
   * (which that chain read) */
  if (r->side_busy) HIP_OK(hipStreamWaitEvent(r->main, r->ev_side, 0));
  on_stream(r->main);
  range_launch(s, S, lo, hi, c, rows);
  HIP_OK(hipEventRecord(r->ev_main, r->main));
  on_stream(r->side);
  check(slb_solver_halo_unpack_multi(s, S, r->left >= 0 ? r->recv2[0] : NULL, r->right >= 0 ? r->recv2[1] : NULL), "halo unpack");
  HIP_OK(hipEventRecord(r->ev_side, r->side));
  r->side_busy = 1;
  on_stream(r->main);
  if (S == 2) slb_solver_double_step_finish(s); else slb_solver_multi_step_finish(s, S);                 /* the output arrays become the current lattices */
}

/* Two loop iterations.  rows_first / rows_second (both or neither): [3][pitch] device
 * buffers that receive the rows av() reads of the main lattice after the first and after
 * the second iteration, at this slab's columns. */
void slb_slabs_double_step(slb_slabs *r, float *rows_first, float *rows_second)
{
  float *rows[2] = {rows_first, rows_second};
  slb_slabs_multi_step(r, 2, rows_first && rows_second ? rows : NULL);
}

/* the main stream waits for the side stream's halo work (before anything but another
 * double step touches the lattices) */
void slb_slabs_join(slb_slabs *r)
{
  if (r->side_busy) {
    HIP_OK(hipStreamWaitEvent(r->main, r->ev_side, 0));
    r->side_busy = 0;
  }
  on_stream(r->main);
}

/* The LAST loop iteration of a run as one fused launch.  A slab forms the main sweep one
 * column into its ghost zone itself, so nothing has to travel -- and nothing does: the
 * ghost columns are stale afterwards, no further step may follow. */
void slb_slabs_last_step(slb_slabs *r)
{
  MARK(r, "slb_slabs_last_step");
  r->single_steps++;
  slb_slabs_join(r);
  slb_solver_step_fused(r->s);
  HIP_OK(hipEventRecord(r->ev_main, r->main));
}

/* One loop iteration as one fused launch, after which any step may follow: the slab forms
 * both sweeps for all its columns (as above), then the ghost columns of both new lattices
 * come from the neighbours -- G-1 columns of the main lattice, G of the half-step one, the
 * depth a launch of three steps reads. */
void slb_slabs_step(slb_slabs *r)
{
  slb_solver *s = r->s;
  MARK(r, "slb_slabs_step");
  r->single_steps++;
  slb_slabs_join(r);
  slb_solver_step_fused(s);
  HIP_OK(hipEventRecord(r->ev_main, r->main));
  HIP_OK(hipStreamWaitEvent(r->side, r->ev_main, 0));
  on_stream(r->side);
  const int which[2] = {3, 0}, ncols[2] = {s->G - 1, s->G};  /* current main lattice, current half-step lattice */
  for (int k = 0; k < 2; k++) {
    for (int side = 0; side < 2; side++)
      if ((side == 0 ? r->left : r->right) >= 0)
        check(slb_solver_halo_pack(s, which[k], side, ncols[k], r->send4[side]), "halo pack");
    exchange(r, r->send4, r->recv4, 2 * (size_t)ncols[k] * (size_t)(s->N + 1));
    for (int side = 0; side < 2; side++)
      if ((side == 0 ? r->left : r->right) >= 0)
        check(slb_solver_halo_unpack(s, which[k], side, ncols[k], r->recv4[side]), "halo unpack");
  }
  HIP_OK(hipEventRecord(r->ev_side, r->side));
  r->side_busy = 1;
  on_stream(r->main);
}

/* ---- av chain --------------------------------------------------------------------------- */
static void batch_reset(slb_slabs *r, av_batch *b)
{
  const size_t row_bytes = sizeof(float) * 3 * (size_t)r->s->pitch;
  if (!b->init) {
    for (int k = 0; k < AVC_SETS; k++) {
      HIP_OK(hipMalloc((void **)&b->rows[k], row_bytes));
      HIP_OK(hipMemset(b->rows[k], 0, row_bytes));
    }
    HIP_OK(hipMalloc((void **)&b->init, sizeof(float) * 3 * AVC_SETS));
    HIP_OK(hipMalloc((void **)&b->tot, sizeof(float) * 3 * AVC_SETS));
    HIP_OK(hipDeviceSynchronize());
  } else {
    /* the buffers' last occupant has been summed before the time loop writes them again */
    HIP_OK(hipStreamWaitEvent(r->main, b->done, 0));
    HIP_OK(hipStreamWaitEvent(r->side, b->done, 0));
  }
  b->n = 0; b->steps = 0; b->ready = -1; b->summed = 0; b->sent = 0;
}

/* Row buffers for the next one to three lattices to average (the next launch's, or the
 * last single step's): rows[0..n-1]. */
void slb_slabs_av_rows(slb_slabs *r, int n, float *rows[])
{
  av_batch *b = batch_at(r, r->batch_open);
  if (b->ready >= 0 || !b->init) {
    if (b->ready >= 0 && !(b->summed && (b->sent || r->right < 0))) {
      fprintf(stderr, "ERROR: av chain: ring of %d batches is too short\n", r->ring_n);
      exit(EXIT_FAILURE);
    }
    batch_reset(r, b);
  }
  for (int k = 0; k < n; k++) rows[k] = b->rows[b->n + k];
}

/* sums of every complete batch whose start values are here, in order */
static void av_launch_ready(slb_slabs *r)
{
  for (;;) {
    if (r->batch_next_sum >= r->batch_open) return;           /* only complete batches */
    av_batch *b = batch_at(r, r->batch_next_sum);
    if (r->left >= 0 && b->ready + (long)AVC_LAG * r->rank >= r->exchanges) return;   /* start values not here yet */
    const float *a[AVC_SETS], *bb[AVC_SETS], *ini[AVC_SETS];
    float *tot[AVC_SETS];
    for (int k = 0; k < b->n; k++) {
      a[k] = b->rows[k]; bb[k] = b->rows[k] + r->s->pitch;      /* b's row 1 is looked for at b + pitch = buffer row 2 */
      ini[k] = r->left >= 0 ? b->init + 3 * k : NULL;
      tot[k] = b->tot + 3 * k;
    }
    /* the rows were written by the interior launch and the edge launches of their double steps */
    HIP_OK(hipStreamWaitEvent(r->av, r->ev_main, 0));
    if (r->side_busy) HIP_OK(hipStreamWaitEvent(r->av, r->ev_side, 0));
    slb_push_stream(r->av);
    check(slb_av_sums_exact_from_n(&r->s->av_slab, b->n, a, bb, ini, tot), "the sums of av()");
    if (r->right < 0)
      for (int k = 0; k < b->n; k++) check(slb_av_update_exact(b->tot + 3 * k, r->av_data, b->t[k]), "the update of av()");
    slb_pop_stream();
    HIP_OK(hipEventRecord(b->done, r->av));
    b->summed = 1;
    r->batch_next_sum++;
  }
}

static void batch_close(slb_slabs *r)
{
  av_batch *b = batch_at(r, r->batch_open);
  if (b->ready >= 0 || b->n == 0) return;
  /* the exchange of the step that completed it; one exchange carries one batch, so a batch
   * completed without an exchange of its own (the last single step) counts as the next one's */
  long ready = r->exchanges - 1 >= 0 ? r->exchanges - 1 : 0;
  if (r->batch_open > 0 && ready <= batch_at(r, r->batch_open - 1)->ready) ready = batch_at(r, r->batch_open - 1)->ready + 1;
  b->ready = ready;
  r->batch_open++;
  if (r->left < 0) av_launch_ready(r);
}

/* The launch that fills the buffers handed out by slb_slabs_av_rows() has been queued: n
 * lattices at loop times t[0..n-1], buffers rows[skip .. skip+n-1] of that call (skip > 0:
 * the first `skip` iterations of the launch are not averaged). */
void slb_slabs_av_submit(slb_slabs *r, int n, const float *t, int skip)
{
  av_batch *b = batch_at(r, r->batch_open);
  if (skip) {                                  /* the lattices to average sit behind `skip` others: bring them forward */
    float *x[AVC_SETS];
    for (int k = 0; k < skip + n; k++) x[k] = b->rows[b->n + k];
    for (int k = 0; k < n; k++) b->rows[b->n + k] = x[skip + k];
    for (int k = 0; k < skip; k++) b->rows[b->n + n + k] = x[k];
  }
  for (int k = 0; k < n; k++) b->t[b->n + k] = t[k];
  b->n += n;
  b->steps++;
  if (b->steps >= AVC_BATCH || b->n + 3 > AVC_SETS) batch_close(r);     /* room for a launch of three */
}

/* the three rows of the current main lattice -> the next row buffer, then as above */
void slb_slabs_av_submit_current(slb_slabs *r, float t)
{
  float *rows[3];
  slb_slabs_av_rows(r, 1, rows);
  slb_slabs_join(r);
  check(slb_solver_copy_av_rows(r->s, rows[0]), "copying the rows of av()");
  HIP_OK(hipEventRecord(r->ev_main, r->main));
  slb_slabs_av_submit(r, 1, &t, 0);
}

/* Everything submitted is summed and applied; av6 (host, 6 floats) receives the averaging
 * data on rank 0 (other ranks: their own copy is meaningless and av6 is left alone). */
void slb_slabs_av_finish(slb_slabs *r, float av6[6])
{
  MARK(r, "slb_slabs_av_finish: draining the chain");
  batch_close(r);
  /* empty exchanges until the last batch has passed the last rank */
  const long last = r->batch_open - 1;
  if (last >= 0) {
    const long until = batch_at(r, last)->ready + (long)AVC_LAG * (r->nranks - 1);
    on_stream(r->side);
    while (r->exchanges <= until) exchange(r, NULL, NULL, 0);
    on_stream(r->main);
  }
  MARK(r, "slb_slabs_av_finish: waiting for the side and av streams");
  HIP_OK(hipStreamSynchronize(r->side));
  HIP_OK(hipStreamSynchronize(r->av));
  MARK(r, "slb_slabs_av_finish: averaging data to rank 0");
  if (r->nranks > 1) {
    /* the last rank holds the result; rank 0 writes the record */
    slb_link_begin(r->link);
    if (r->rank == r->nranks - 1) slb_link_send(r->link, 0, r->av_data, 6, r->av);
    if (r->rank == 0) slb_link_recv(r->link, r->nranks - 1, r->av_data, 6, r->av);
    slb_link_end(r->link, r->av);
    HIP_OK(hipStreamSynchronize(r->av));
  }
  if (r->rank == 0) HIP_OK(hipMemcpy(av6, r->av_data, 6 * sizeof(float), hipMemcpyDeviceToHost));
}

/* The averaging data as rank 0 has them on the host become the chain's (they live on the
 * last rank): display=77 rescales them in place at every record
 * (reference: src/boltzmann_c_solver.c:319-325). */
void slb_slabs_av_set(slb_slabs *r, const float av6[6])
{
  HIP_OK(hipStreamSynchronize(r->av));
  if (r->rank == 0) HIP_OK(hipMemcpy(r->av_data, av6, 6 * sizeof(float), hipMemcpyHostToDevice));
  if (r->nranks > 1) {
    slb_link_begin(r->link);
    if (r->rank == 0) slb_link_send(r->link, r->nranks - 1, r->av_data, 6, r->av);
    if (r->rank == r->nranks - 1) slb_link_recv(r->link, 0, r->av_data, 6, r->av);
    slb_link_end(r->link, r->av);
    HIP_OK(hipStreamSynchronize(r->av));
  }
}

/* Rows [row0, row0+nrows) of device array `arr` (numbering of slb_solver_get_rows) over
 * all slabs: on rank 0, dst receives nrows x (M+3) floats (global columns 0..M+2); other
 * ranks send their columns and leave dst alone. */
void slb_slabs_gather_rows(slb_slabs *r, int arr, int row0, int nrows, float *dst)
{
  slb_solver *s = r->s;
  MARK(r, "slb_slabs_gather_rows: waiting for the main and side streams");
  slb_slabs_join(r);
  HIP_OK(hipStreamSynchronize(r->main));
  HIP_OK(hipStreamSynchronize(r->side));
  MARK(r, "slb_slabs_gather_rows: row messages");
  const float *src = arr == 0 ? s->a0 : (arr <= 4 ? s->a[arr - 1] : s->b[arr - 5]);
  const int total = s->M + 3;
  /* my columns: the owned ones, plus the global boundary columns on the outer slabs */
  const int lo = r->rank == 0 ? s->G - 1 : s->G;
  const int hi = r->rank == r->nranks - 1 ? s->G + s->W + 2 : s->G + s->W;     /* exclusive */
  if (r->nranks == 1) {
    HIP_OK(hipMemcpy2D(dst, sizeof(float) * total, src + (size_t)row0 * s->pitch + lo, sizeof(float) * s->pitch,
                       sizeof(float) * (hi - lo), nrows, hipMemcpyDeviceToHost));
    return;
  }
  float *all = NULL;
  if (r->rank == 0) HIP_OK(hipMalloc((void **)&all, sizeof(float) * (size_t)nrows * total));
  /* every rank can work out every other rank's columns; one message per row and rank */
  const int base = s->M / r->nranks, rem = s->M % r->nranks;
  if (r->rank == 0) {
    for (int q = 1; q < r->nranks; q++) {
      const int first = 1 + q * base + (q < rem ? q : rem), w = base + (q < rem ? 1 : 0);
      const int n = q == r->nranks - 1 ? w + 2 : w;
      for (int k = 0; k < nrows; k++) {
        slb_link_begin(r->link);
        slb_link_recv(r->link, q, all + (size_t)k * total + first, (size_t)n, r->main);
        slb_link_end(r->link, r->main);
      }
    }
    HIP_OK(hipMemcpy2DAsync(all, sizeof(float) * total, src + (size_t)row0 * s->pitch + lo, sizeof(float) * s->pitch,
                            sizeof(float) * (hi - lo), nrows, hipMemcpyDeviceToDevice, r->main));
    HIP_OK(hipStreamSynchronize(r->main));
    HIP_OK(hipMemcpy(dst, all, sizeof(float) * (size_t)nrows * total, hipMemcpyDeviceToHost));
    HIP_OK(hipFree(all));
  } else {
    for (int k = 0; k < nrows; k++) {
      slb_link_begin(r->link);
      slb_link_send(r->link, 0, src + (size_t)(row0 + k) * s->pitch + lo, (size_t)(hi - lo), r->main);
      slb_link_end(r->link, r->main);
    }
    HIP_OK(hipStreamSynchronize(r->main));
  }
}

/* The values f(phi_x, phi_y) of the current main lattice for every phi_x block of a frame
 * (reference: src/boltzmann_solver.c:334-346, 487-507): every slab forms the values of its own
 * columns on its GPU, a batch of phi_x values at a time; rank 0 puts the rows together and
 * hands each block -- the M+1 values of columns 1..M+1, and with_a0 the equilibrium
 * distribution's as well -- to `line` (p = number of the block). */
typedef void (*frame_block_fn)(void *ctx, int p, float phi_x, const float *vals, const float *vals0);

static void frame_blocks(slb_slabs *r, int with_a0, frame_block_fn line, void *ctx)
{
  slb_solver *s = r->s;
  MARK(r, "frame_blocks: waiting for the main and side streams");
  slb_slabs_join(r);
  HIP_OK(hipStreamSynchronize(r->main));
  HIP_OK(hipStreamSynchronize(r->side));
  MARK(r, "frame_blocks: frame values and their messages");
  const int m1 = s->M + 1, sets = with_a0 ? 2 : 1;
  float phis[1024];
  const int nphi = slb_frame_phi_x(phis, 1024);
  if (nphi > 1024) { fprintf(stderr, "ERROR: more than 1024 phi_x blocks\n"); exit(EXIT_FAILURE); }
  int batch = (int)((16u << 20) / (unsigned)m1);
  batch = batch < 8 ? 8 : (batch > 640 ? 640 : batch);
  const int mine = s->main_slab.j_last - s->main_slab.j_first + 1;
  const int base = s->M / r->nranks, rem = s->M % r->nranks;
  float *d_mine = NULL, *d_all = NULL, *h_all = NULL, *line_v = NULL;
  HIP_OK(hipMalloc((void **)&d_mine, sizeof(float) * (size_t)batch * mine));
  if (r->rank == 0) {
    HIP_OK(hipMalloc((void **)&d_all, sizeof(float) * (size_t)batch * m1));
    h_all = (float *)malloc(sizeof(float) * (size_t)batch * m1 * sets);
    line_v = (float *)malloc(sizeof(float) * (size_t)m1 * sets);
    if (!h_all || !line_v) { fprintf(stderr, "ERROR: out of host memory for the frame\n"); exit(EXIT_FAILURE); }
  }
  on_stream(r->main);
  for (int p0 = 0; p0 < nphi; p0 += batch) {
    const int np = nphi - p0 < batch ? nphi - p0 : batch;
    for (int set = 0; set < sets; set++) {
      check(slb_solver_frame_values_device(s, set, phis + p0, np, d_mine), "the frame values");
      if (r->rank != 0) {
        slb_link_begin(r->link);
        slb_link_send(r->link, 0, d_mine, (size_t)np * mine, r->main);
        slb_link_end(r->link, r->main);
        HIP_OK(hipStreamSynchronize(r->main));
        continue;
      }
      /* rank q's block: np rows of its own columns, at offset np * (columns of the ranks before it) */
      HIP_OK(hipMemcpyAsync(d_all, d_mine, sizeof(float) * (size_t)np * mine, hipMemcpyDeviceToDevice, r->main));
      size_t off = (size_t)np * mine;
      for (int q = 1; q < r->nranks; q++) {
        const int w = base + (q < rem ? 1 : 0) + (q == r->nranks - 1 ? 1 : 0);
        slb_link_begin(r->link);
        slb_link_recv(r->link, q, d_all + off, (size_t)np * w, r->main);
        slb_link_end(r->link, r->main);
        off += (size_t)np * w;
      }
      HIP_OK(hipStreamSynchronize(r->main));
      HIP_OK(hipMemcpy(h_all + (size_t)set * batch * m1, d_all, sizeof(float) * (size_t)np * m1, hipMemcpyDeviceToHost));
    }
    if (r->rank != 0) continue;
    for (int p = 0; p < np; p++) {
      for (int set = 0; set < sets; set++) {
        const float *h = h_all + (size_t)set * batch * m1;
        size_t blk = 0;
        int col = 0;
        for (int q = 0; q < r->nranks; q++) {
          const int w = q == 0 ? mine : base + (q < rem ? 1 : 0) + (q == r->nranks - 1 ? 1 : 0);
          memcpy(line_v + (size_t)set * m1 + col, h + blk + (size_t)p * w, sizeof(float) * (size_t)w);
          blk += (size_t)np * w;
          col += w;
        }
      }
      line(ctx, p0 + p, phis[p0 + p], line_v, with_a0 ? line_v + m1 : NULL);
    }
  }
  if (r->rank == 0) {
    free(h_all); free(line_v);
    HIP_OK(hipFree(d_all));
  }
  HIP_OK(hipFree(d_mine));
}

static void block_frame(void *ctx, int p, float phi_x, const float *vals, const float *vals0)
{
  (void)p; (void)vals0;
  slb_frame_block(ctx, phi_x, vals);
}

/* a display=7 / display=8 frame file of the current main lattice */
int slb_slabs_write_frame(slb_slabs *r, const char *name, float t)
{
  slb_solver *s = r->s;
  const int total = s->M + 3;
  float *a01 = r->rank == 0 ? (float *)malloc(sizeof(float) * 2 * (size_t)total) : NULL;
  slb_slabs_gather_rows(r, 1 + s->current, 0, 2, a01);
  void *f = NULL;
  if (r->rank == 0 && !(f = slb_frame_begin(name, t))) exit(EXIT_FAILURE);
  frame_blocks(r, 0, block_frame, f);
  if (r->rank == 0) {
    slb_frame_end(f, a01, total);
    free(a01);
  }
  return 0;
}

/* a display=9 strobe file: this period's frame added to the running image on rank 0
 * (reference: src/boltzmann_solver.c:459-484) */
struct strobe_ctx { FILE *f; float *image; };
static void block_strobe(void *ctx, int p, float phi_x, const float *vals, const float *vals0)
{
  struct strobe_ctx *c = (struct strobe_ctx *)ctx;
  (void)vals0;
  slb_strobe_block(c->f, c->image, p, phi_x, vals);
}

static void write_strobe(slb_slabs *r, const char *name, float *image, float t)
{
  slb_solver *s = r->s;
  float *a0row = r->rank == 0 ? (float *)malloc(sizeof(float) * ((size_t)s->M + 3)) : NULL;
  slb_slabs_gather_rows(r, 1 + s->current, 0, 1, a0row);
  struct strobe_ctx c = {NULL, image};
  if (r->rank == 0) {
    if (!(c.f = fopen(name, "w"))) { perror("ERROR"); exit(EXIT_FAILURE); }
    printf("\nWriting strobe %s\n", name);
  }
  frame_blocks(r, 0, block_strobe, &c);
  if (r->rank == 0) {
    slb_strobe_end(c.f, a0row, t);
    fclose(c.f);
    free(a0row);
  }
}

static void block_display3(void *ctx, int p, float phi_x, const float *vals, const float *vals0)
{
  (void)p;
  slb_display3_block(ctx, phi_x, vals, vals0);
}

int slb_slabs_ranks_seen(slb_slabs *r) { return r->link ? slb_link_ranks_seen(r->link) : 1; }

void slb_slabs_destroy(slb_slabs *r)
{
  if (!r) return;
  MARK(r, "slb_slabs_destroy: hipDeviceSynchronize");
  HIP_OK(hipDeviceSynchronize());
  if (r->dog_seconds > 0) { r->dog_stop = 1; pthread_join(r->dog, NULL); }
  for (int k = 0; k < r->ring_n; k++) {
    for (int q = 0; q < AVC_SETS; q++) if (r->ring[k].rows[q]) hipFree(r->ring[k].rows[q]);
    if (r->ring[k].init) { hipFree(r->ring[k].init); hipFree(r->ring[k].tot); }
    hipEventDestroy(r->ring[k].done);
  }
  free(r->ring);
  for (int side = 0; side < 2; side++) {
    hipFree(r->send2[side]); hipFree(r->recv2[side]); hipFree(r->send4[side]); hipFree(r->recv4[side]);
  }
  hipFree(r->av_data);
  hipEventDestroy(r->ev_main); hipEventDestroy(r->ev_side); hipEventDestroy(r->ev_x);
  hipStreamDestroy(r->main); hipStreamDestroy(r->side); hipStreamDestroy(r->av);
  slb_solver_destroy(r->s);
  free(r);
}

/* ---- the program over slabs ----------------------------------------------------------------
 * One process per rank, forked before anything has touched a GPU.  Rank q takes device
 * `device` + q when the machine has that many, and the ranks talk over RCCL; with fewer
 * devices (a one-GPU box) the ranks share them and talk over sockets -- a rehearsal that
 * runs the same loop.  SLB_TRANSPORT=fds forces the sockets. */
/* What follows the step taken at loop time t with this frame_time: 0 nothing, 1 only av(),
 * 2 a display=77 record, a display=7 frame or a display=9 strobe test.  Two steps of kind 0 or 1
 * may share a launch. */
static int step_kind(float t, float frame_time)
{
  if (display == 77) return host_E_omega > 0 && frame_time >= 0.01 ? 2 : 0;
  if (display == 7)  return frame_time >= 0.01 && t > frame_start ? 2 : 0;
  if (display == 8)  return 0;
  if (display == 9 && t >= t_start) return 2;
  return host_E_omega > 0 && t >= t_start ? 1 : 0;
}

static int rank_main(int rank, int nranks, const int *fds)
{
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (ndev < 1) { fprintf(stderr, "ERROR: no GPU\n"); return EXIT_FAILURE; }
  const char *tr = getenv("SLB_TRANSPORT");
  const int use_rccl = nranks > 1 && !(tr && strcmp(tr, "fds") == 0) && ndev >= device + nranks;
  HIP_OK(hipSetDevice(use_rccl ? device + rank : (device + rank) % ndev));

  float T = host_omega > 0 ? (2 * SLB_PI / host_omega) : 0;
  float *strobe = NULL;
  t_max = display == 9 ? t_start + 101 * T : t_start + T;      /* display=9: 101 periods, one strobe image each */
  if (display == 9 && rank == 0 && !(strobe = slb_strobe_image())) return EXIT_FAILURE;
  if (rank == 0 && quiet == 0) printf("# t_max = %0.20f\n", t_max);
  slb_derive_and_load();

  slb_link *link = NULL;
  if (nranks > 1 && use_rccl) {
    char id[128];
    if (rank == 0) {
      if (slb_link_unique_id(id) != 0) return EXIT_FAILURE;
      for (int q = 1; q < nranks; q++)
        if (write(fds[q], id, sizeof(id)) != (ssize_t)sizeof(id)) { perror("ERROR"); return EXIT_FAILURE; }
    } else if (read(fds[0], id, sizeof(id)) != (ssize_t)sizeof(id)) { perror("ERROR"); return EXIT_FAILURE; }
    link = slb_link_open_rccl(rank, nranks, id);
  } else if (nranks > 1) {
    link = slb_link_open_fds(rank, nranks, fds);
  }
  if (nranks > 1 && !link) return EXIT_FAILURE;
  if (rank == 0 && nranks > 1 && quiet == 0)
    printf("# %d phi_y slabs, messages over %s (%d ranks seen)\n", nranks, slb_link_transport(link),
           slb_link_ranks_seen(link));

  slb_slabs *r = slb_slabs_create(rank, nranks, link);
  if (!r) return EXIT_FAILURE;
  slb_solver *s = r->s;
  slb_set_av_mode(SLB_AV_EXACT);
  slb_slabs_tiptoe(r);
  /* fault injection for the tests of the parent's error path: this rank ends like any of the
   * exit(EXIT_FAILURE) paths would, while its neighbours wait for its next message */
  const char *fail = getenv("SLB_FAIL_RANK");
  if (fail && nranks > 1 && atoi(fail) == rank) {
    fprintf(stderr, "ERROR: rank %d ends here (SLB_FAIL_RANK)\n", rank);
    exit(EXIT_FAILURE);
  }

  const int timing = getenv("SLB_TIMING") != NULL;
  const char *form = getenv("SLB_FORM");
  const int triples = !(form != NULL && strcmp(form, "double") == 0) && slb_solver_steps_per_launch(s) == 3;
  int progress = 0, frame_number = 1;
  float t = 0, t0 = 0, frame_time = 0, last_phase = 0;
  char file_name[64];
  const int total = host_M + 3;
  float *a012 = rank == 0 ? (float *)malloc(sizeof(float) * 3 * (size_t)total) : NULL;
  float *b12  = rank == 0 ? (float *)malloc(sizeof(float) * 2 * (size_t)total) : NULL;
  for (;;) {
    struct timespec w0, w1;
    long loop_steps = 0;
    if (timing) { HIP_OK(hipDeviceSynchronize()); clock_gettime(CLOCK_MONOTONIC, &w0); }
    for (t = t0; t < t_max; t += host_dt, loop_steps++) {
      const int k1 = step_kind(t, frame_time);
      const float t2 = t + host_dt, ft2 = frame_time + host_dt;
      const int k2 = step_kind(t2, ft2);
      const float t3 = t2 + host_dt, ft3 = ft2 + host_dt;
      const int k3 = step_kind(t3, ft3);
      s->t = t;                              /* the engine forms the cosines from its own clock */
      if (triples && t2 < t_max && t3 < t_max && k1 < 2 && k2 < 2 && k3 < 2) {
        /* three iterations in one launch */
        const int n_av = k1 + k2 + k3;
        if (n_av) {
          float *rows[3];
          slb_slabs_av_rows(r, 3, rows);
          slb_slabs_multi_step(r, 3, rows);
          const float tk[3] = {t, t2, t3};
          slb_slabs_av_submit(r, n_av, tk + (3 - n_av), 3 - n_av);   /* averaging, once begun, goes on */
        } else {
          slb_slabs_multi_step(r, 3, NULL);
        }
        t = t3;
        loop_steps += 2;
        frame_time = ft3 + host_dt;
        if (rank == 0 && out != stdout && display != 7 && quiet == 0 && (progress += 3) >= 300) {
          printf("\rt=%0.9f %0.2f%%", t, t / t_max * 100);
          fflush(stdout);
          progress = 0;
        }
        continue;
      }
      if (t2 < t_max && k1 < 2 && k2 < 2) {
        /* two iterations in one launch: same bits, the lattice in between is never stored
         * (the rows av() reads of it are) */
        if (k1 || k2) {
          float *rows[2];
          slb_slabs_av_rows(r, 2, rows);
          slb_slabs_double_step(r, rows[0], rows[1]);
          const float tt[2] = {k1 ? t : t2, t2};
          slb_slabs_av_submit(r, k1 + k2, tt, k1 ? 0 : 1);
        } else {
          slb_slabs_double_step(r, NULL, NULL);
        }
        t = t2;
        loop_steps++;
        frame_time = ft2 + host_dt;
        if (rank == 0 && out != stdout && display != 7 && quiet == 0 && (progress += 2) >= 300) {
          printf("\rt=%0.9f %0.2f%%", t, t / t_max * 100);
          fflush(stdout);
          progress = 0;
        }
        continue;
      }
      /* the last iteration of the run lets nothing travel; with read-from= another pass may follow */
      if (t2 < t_max || read_from != NULL) slb_slabs_step(r); else slb_slabs_last_step(r);
      if (display == 77 && k1 == 2) {
        /* a display=77 record: av() of the lattice this step wrote, the sums from the one
         * before it (reference: src/boltzmann_c_solver.c:180-187, 298-331) */
        float av6[6] = {0, 0, 0, 0, 0, 0};
        slb_slabs_av_submit_current(r, t);
        slb_slabs_av_finish(r, av6);
        slb_slabs_gather_rows(r, 1 + s->next, 0, 3, a012);     /* after the swap `next` is the old lattice */
        slb_slabs_gather_rows(r, 5 + s->next, 1, 2, b12);
        if (rank == 0) {
          const float norm = slb_host_norm(a012, host_M, host_dPhi, host_alpha);
          slb_write_display77(norm, a012, b12, total, av6, t);
        }
        slb_slabs_av_set(r, av6);            /* the CPU solver rescales its averaging data in place */
        frame_time = 0;
      } else if (k1 == 1) {
        slb_slabs_av_submit_current(r, t);     /* (display=9 prints no averages: its steps skip av()) */
      }
      if (display == 9 && t >= t_start) {
        /* a strobe image whenever t/T wraps (reference: src/boltzmann_solver.c:260-275) */
        const float periods = t / T;
        const float phase = periods - ((int)periods);      /* float arithmetic, as the single-GPU loop has it */
        if (phase < last_phase) {
          sprintf(file_name, "strobe%08d.data", frame_number++);
          write_strobe(r, file_name, strobe, t);
          frame_time = 0;
        }
        last_phase = phase;
      }
      if (display == 7 && k1 == 2) {                                  /* a movie frame */
        sprintf(file_name, "frame%08d.data", frame_number++);
        slb_slabs_write_frame(r, file_name, t);
        frame_time = 0;
      }
      if (rank == 0 && out != stdout && display != 7 && quiet == 0 && ++progress >= 300) {
        printf("\rt=%0.9f %0.2f%%", t, t / t_max * 100);
        fflush(stdout);
        progress = 0;
      }
      frame_time += host_dt;
    }
    float av6[6] = {0, 0, 0, 0, 0, 0};
    slb_slabs_av_finish(r, av6);
    if (timing) {
      HIP_OK(hipDeviceSynchronize());
      clock_gettime(CLOCK_MONOTONIC, &w1);
      const double sec = (double)(w1.tv_sec - w0.tv_sec) + 1e-9 * (double)(w1.tv_nsec - w0.tv_nsec);
      if (rank == 0)
        fprintf(stderr, "# time loop: %ld iterations in %.4f s (%.4f ms each)\n", loop_steps, sec,
                loop_steps ? 1e3 * sec / loop_steps : 0.0);
    }

    /* the three rows the closing sums read, put together on rank 0 */
    slb_slabs_gather_rows(r, 1 + s->current, 0, 2, a012);
    slb_slabs_gather_rows(r, 5 + s->current, 1, 1, b12);
    const float norm = rank == 0 ? slb_host_norm(a012, host_M, host_dPhi, host_alpha) : 0;
    if (display == 3) {
      /* the final distribution next to the equilibrium one
       * (reference: src/boltzmann_c_solver.c:219-234) */
      frame_blocks(r, 1, block_display3, out);
      if (rank == 0) {
        fprintf(out, "# norm=%0.20f\n", norm);
        printf("# norm=%0.20f\n", norm);
      }
      break;
    }
    if (display == 8) { slb_slabs_write_frame(r, "frame.data", t); break; }
    if (display == 4 && rank == 0) slb_write_display4(norm, a012, b12, total, av6, T);
    if (read_from == NULL) break;

    /* new parameters: rank 0 owns the input and tells the others (over the start-up sockets) */
    float msg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (rank == 0) {
      msg[7] = (float)slb_read_new_parameters(msg);
      for (int q = 1; q < nranks; q++)
        if (write(fds[q], msg, sizeof(msg)) != (ssize_t)sizeof(msg)) { perror("ERROR"); return EXIT_FAILURE; }
    } else if (read(fds[0], msg, sizeof(msg)) != (ssize_t)sizeof(msg)) { perror("ERROR"); return EXIT_FAILURE; }
    if (msg[7] == 0) break;
    slb_apply_new_parameters(msg, t, &T, &t0);
    const float zero6[6] = {0, 0, 0, 0, 0, 0};
    slb_slabs_av_set(r, zero6);
    if (rank == 0 && quiet == 0) printf("# t_max = %0.20f\n", t_max);
  }
  if (rank == 0) slb_close_output();
  free(a012); free(b12); free(strobe);
  slb_slabs_destroy(r);
  slb_link_close(link);
  return EXIT_SUCCESS;
}

int slb_slab_main(int nranks)
{
  if (display == 77 && host_N < 2) { fprintf(stderr, "ERROR: display=77 needs n-harmonics >= 2.\n"); return EXIT_FAILURE; }
  if (nranks < 1 || nranks > 64) { fprintf(stderr, "ERROR: Invalid value of ranks= parameter.\n"); return EXIT_FAILURE; }
  if (nranks == 1) return rank_main(0, 1, NULL);
  /* a socket for every pair of ranks: start-up data, and the messages themselves when the
   * ranks share a GPU */
  int (*pair)[2] = (int (*)[2])malloc(sizeof(int[2]) * (size_t)nranks * nranks);
  for (int i = 0; i < nranks; i++)
    for (int j = i + 1; j < nranks; j++)
      if (socketpair(AF_UNIX, SOCK_STREAM, 0, pair[i * nranks + j]) != 0) { perror("ERROR"); return EXIT_FAILURE; }
  fflush(NULL);
  pid_t *pids = (pid_t *)malloc(sizeof(pid_t) * (size_t)nranks);
  for (int q = 0; q < nranks; q++) {
    pids[q] = fork();
    if (pids[q] < 0) { perror("ERROR"); return EXIT_FAILURE; }
    if (pids[q] == 0) {
      int *fds = (int *)malloc(sizeof(int) * (size_t)nranks);
      for (int i = 0; i < nranks; i++)
        for (int j = i + 1; j < nranks; j++) {
          int *p = pair[i * nranks + j];
          if (i == q)      { fds[j] = p[0]; close(p[1]); }
          else if (j == q) { fds[i] = p[1]; close(p[0]); }
          else             { close(p[0]); close(p[1]); }
        }
      fds[q] = -1;
      if (q != 0) out = NULL;                 /* rank 0 writes the record */
      const int rc = rank_main(q, nranks, fds);
      fflush(NULL);
      _exit(rc);
    }
  }
  for (int i = 0; i < nranks; i++)
    for (int j = i + 1; j < nranks; j++) { close(pair[i * nranks + j][0]); close(pair[i * nranks + j][1]); }
  /* Wait for whichever rank ends next.  A rank that fails leaves its neighbours blocked in a
   * receive for good, so on the first abnormal end the others are ended too and the run
   * returns failure instead of hanging. */
  int worst = EXIT_SUCCESS, alive = nranks;
  while (alive > 0) {
    int st = 0;
    const pid_t who = waitpid(-1, &st, 0);
    if (who < 0) { if (errno == EINTR) continue; worst = EXIT_FAILURE; break; }
    int q = 0;
    while (q < nranks && pids[q] != who) q++;
    if (q == nranks) continue;                  /* not one of the ranks */
    pids[q] = -1;
    alive--;
    if (WIFEXITED(st) && WEXITSTATUS(st) == 0) continue;
    if (worst == EXIT_SUCCESS) {
      if (WIFSIGNALED(st)) fprintf(stderr, "ERROR: rank %d was ended by signal %d; ending the other ranks\n", q, WTERMSIG(st));
      else                 fprintf(stderr, "ERROR: rank %d ended with status %d; ending the other ranks\n", q, WEXITSTATUS(st));
      for (int k = 0; k < nranks; k++) if (pids[k] > 0) kill(pids[k], SIGTERM);
    }
    worst = EXIT_FAILURE;
  }
  free(pids); free(pair);
  return worst;
}
