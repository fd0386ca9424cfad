/*
 * slb_slabs.h -- the loop over several GPUs as C entry points: one process per GPU, each
 * with a phi_y slab of the lattice; halo columns and the running sums of av() travel
 * between neighbouring processes over RCCL, called from C (the library is opened at run
 * time, a single-GPU program never loads it).
 *
 * No counterpart in the reference, which runs on the one GPU its device= option selects
 * (reference: src/boltzmann_solver.c:77); what these calls replace is the body of its time
 * loop (src/boltzmann_solver.c:196-302).  Exported by libslb2d.so next to the boundary of
 * slb_abi.h; load_data() -- or the host engine's slb_derive_and_load() -- must have run.
 *
 * A run:   rank 0: slb_link_unique_id(id) and hands id to the others (any channel);
 *          every rank, its GPU current: link = slb_link_open_rccl(rank, nranks, id);
 *          run = slb_slabs_create(rank, nranks, link); slb_slabs_tiptoe(run);
 *          slb_slabs_multi_step(run, 3, NULL) ...; slb_slabs_last_step(run);
 *          slb_slabs_gather_rows(...) on every rank (rank 0 receives);
 *          slb_slabs_destroy(run); slb_link_close(link).
 * Every rank makes the same calls in the same order: messages are matched by order.
 */
#ifndef SLB_SLABS_H
#define SLB_SLABS_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct slb_solver slb_solver;      /* the engine object of one slab (host/slb_host.h) */

/* Point-to-point messages between the processes of a run (host/slb_link.c): device buffers
 * of floats, sent and received in groups on a HIP stream.  RCCL (ncclSend/ncclRecv over
 * xGMI; the library is opened at run time) or, for rehearsals with several ranks on one
 * GPU, a socket per peer with the messages staged through host memory. */
typedef struct slb_link slb_link;
int       slb_link_unique_id(char unique_id[128]);            /* ncclGetUniqueId, for rank 0 to hand out */
slb_link *slb_link_open_rccl(int rank, int nranks, const char unique_id[128]);   /* current device = this rank's */
slb_link *slb_link_open_fds(int rank, int nranks, const int *fds);               /* fds[peer], -1 for self */
int       slb_link_rank(const slb_link *l);
int       slb_link_nranks(const slb_link *l);
int       slb_link_ranks_seen(const slb_link *l);             /* what the transport itself reports (ncclCommCount) */
const char *slb_link_transport(const slb_link *l);
void      slb_link_begin(slb_link *l);
void      slb_link_send(slb_link *l, int peer, const float *dev, size_t n, void *hip_stream);
void      slb_link_recv(slb_link *l, int peer, float *dev, size_t n, void *hip_stream);
void      slb_link_end(slb_link *l, void *hip_stream);
void      slb_link_close(slb_link *l);

/* The time loop of one slab with its halo traffic and its share of av() (host/slb_slabs.c).
 * link may be NULL for a single rank.  The launches go to streams of the object's own;
 * slb_slabs_main_stream() is the one the interior launches use. */
typedef struct slb_slabs slb_slabs;
slb_slabs  *slb_slabs_create(int rank, int nranks, slb_link *link);
slb_slabs  *slb_slabs_create_loopback(int rank, int nranks, slb_link *link);   /* timing aid, see slb_slabs.c */
void        slb_slabs_destroy(slb_slabs *r);
slb_solver *slb_slabs_solver(slb_slabs *r);
void       *slb_slabs_main_stream(slb_slabs *r);
int         slb_slabs_ranks_seen(slb_slabs *r);
void        slb_slabs_tiptoe(slb_slabs *r);
void        slb_slabs_double_step(slb_slabs *r, float *rows_first, float *rows_second);
/* S = 2 or 3 iterations per launch (3: the form a wide slab runs fastest); rows: NULL, or S
 * [3][pitch] device buffers for the rows av() reads of the S new main lattices */
void        slb_slabs_multi_step(slb_slabs *r, int S, float *const *rows);
void        slb_slabs_step(slb_slabs *r);          /* one iteration, ghost columns refreshed */
void        slb_slabs_last_step(slb_slabs *r);     /* the last iteration of a run: nothing travels */
void        slb_slabs_join(slb_slabs *r);
/* av() over slabs, bit-identical to the single-GPU av(): buffers for the next step's rows,
 * the step's loop times once it has been queued, and the end of the chain (av6 on rank 0) */
void        slb_slabs_av_rows(slb_slabs *r, int n, float *rows[]);    /* n = 1..3 */
void        slb_slabs_av_submit(slb_slabs *r, int n, const float *t, int skip);
void        slb_slabs_av_submit_current(slb_slabs *r, float t);
void        slb_slabs_av_finish(slb_slabs *r, float av6[6]);
void        slb_slabs_av_set(slb_slabs *r, const float av6[6]);
float      *slb_slabs_av_data(slb_slabs *r);
void        slb_slabs_gather_rows(slb_slabs *r, int arr, int row0, int nrows, float *dst);
int         slb_slabs_write_frame(slb_slabs *r, const char *name, float t);   /* a display=7 / display=8 frame over slabs */

#ifdef __cplusplus
}
#endif
#endif
