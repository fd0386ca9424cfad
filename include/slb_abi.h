/*
 * slb_abi.h -- C boundary of libslb2d.so (MI355X / gfx950 build of the
 * super-lattice Boltzmann stepping loop).
 *
 * Part 1 re-exports, symbol for symbol, the boundary the reference's host
 * program binds (declarations: reference src/boltzmann_gpu.h:4-29).  Part 2 is
 * additive: the same kernels with the slab geometry spelled out (for phi_y
 * partitioning across GPUs), halo movers, the reductions split into a
 * partial-sum and an update stage, and launch controls.  Everything is plain C:
 * raw device pointers, ints and floats; no C++ or torch types.
 *
 * All pointers named a*, b*, av_data, sums, buf are DEVICE pointers owned by the
 * caller.  The library keeps no pointer between calls and allocates nothing
 * per call (one small workspace is allocated on first use of av() and
 * slb_av_partial()).
 * Launches are asynchronous on the stream set by slb_set_stream() (default: the
 * null stream, as in the reference), or on the one the calling thread has pushed
 * with slb_push_stream().
 *
 * Like the reference's boundary the library is meant for ONE host thread per
 * process and one GPU per process: load_data(), slb_set_stream(), slb_set_av_mode()
 * and slb_set_tuning() set process-wide state that the following launches read.
 */
#ifndef SLB_ABI_H
#define SLB_ABI_H

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------
 * Part 1 -- the reference boundary
 * ---------------------------------------------------------------------- */

/* Host-owned parameters the boundary imports (reference:
 * src/boltzmann_gpu.cu:40-44; defined by the host in src/boltzmann_cli.c:26-45
 * and src/boltzmann_solver.c:47-65).  libslb2d.so carries the definitions so a
 * foreign-function loader can set them; a host that links the library sets
 * them exactly as the reference's main() does and then calls load_data(). */
extern float host_E_dc, host_E_omega, host_omega, host_mu, host_alpha,
             PhiYmin, PhiYmax, host_B, t_start, host_dPhi, host_dt,
             host_bdt, host_nu_tilde, host_nu2, host_nu;
extern int   host_M, host_N, MSIZE, MP1, NSIZE, host_TMSIZE, PADDED_MSIZE;

/* Snapshot the host_* parameters for the following launches.  Re-callable at
 * any time (reference: src/boltzmann_gpu.cu:57-82, called at
 * src/boltzmann_solver.c:117 and again at :391). */
void load_data(void);

/* Main-lattice sweep: rows n=0..N-1, columns m=1..M+1.  centre = *_current,
 * stencil = *_current_hs, output = *_next.  `blocks`, `t`, `t_hs` are accepted
 * and ignored, as every reference kernel body ignores t and t_hs.
 * (reference: src/boltzmann_gpu.cu:1169-1216; arithmetic and column range:
 * src/boltzmann_c_solver.c:355-382) */
void step_on_grid(int blocks, float *a0, float *a_current, float *b_current,
                  float *a_next, float *b_next,
                  float *a_current_hs, float *b_current_hs,
                  float t, float t_hs, float cos_omega_t, float cos_omega_t_plus_dt);

/* Half-step-lattice sweep: rows n=0..N-1, columns m=1..M.  centre =
 * *_current_hs, stencil = *_next, output = *_next_hs.
 * (reference: src/boltzmann_gpu.cu:1218-1265; arithmetic and column range:
 * src/boltzmann_c_solver.c:384-411 -- the CPU range, which stops one column
 * before the reference's CUDA kernels do) */
void step_on_half_grid(int blocks, float *a0, float *a_current, float *b_current,
                       float *a_next, float *b_next,
                       float *a_current_hs, float *b_current_hs,
                       float *a_next_hs, float *b_next_hs,
                       float t, float t_hs, float cos_omega_t, float cos_omega_t_plus_dt);

/* Harmonic reductions over m=1..M and the running averages / absorption
 * integrals in av_data[0..5] (device, read-modify-write).
 * (reference: src/boltzmann_gpu.cu:1267-1271; arithmetic:
 * src/boltzmann_c_solver.c:413-437) */
void av(int blocks, float *a, float *b, float *av_data, float t);

/* Print "<error string> in <file> at line <line>" and exit(EXIT_FAILURE) when
 * err is not hipSuccess (reference: src/boltzmann_gpu.cu:29-36; err is a
 * hipError_t passed as int). */
void HandleError(int err, const char *file, int line);

/* ------------------------------------------------------------------------
 * Part 2 -- additive entry points (no counterpart in the reference)
 * ---------------------------------------------------------------------- */

/* Geometry of one phi_y slab.  A single-GPU run is the slab
 * {pitch=PADDED_MSIZE, m_offset=0, j_first=1, j_last=M+1 (or M)}. */
typedef struct slb_slab {
  int pitch;     /* floats per row of every array passed with this slab        */
  int m_offset;  /* global column index of local column 0                      */
  int j_first;   /* first local column the sweep writes                        */
  int j_last;    /* last local column the sweep writes (inclusive)             */
} slb_slab;

/* The sweep both reference entry points reduce to: update rows 0..N-1, local
 * columns j_first..j_last of a slab, reading the stencil arrays at local
 * columns j-1 and j+1.  Physics parameters come from the last load_data().
 * Returns 0, or a hipError_t value if the launch was rejected; argument
 * errors (bad geometry, misaligned arrays for the chosen variant) return -1
 * and print the reason. */
int slb_sweep(const slb_slab *slab,
              const float *a0, const float *a_centre, const float *b_centre,
              const float *a_stencil, const float *b_stencil,
              float *a_out, float *b_out,
              float cos_omega_t, float cos_omega_t_plus_dt);

/* Geometry of a fused step (both sweeps in one launch) on one slab. */
typedef struct slb_step_geom {
  int pitch;                /* floats per row, a multiple of 4                        */
  int m_offset;             /* global column index of local column 0                  */
  int j1_first, j1_last;    /* local columns the main-lattice sweep writes            */
  int j2_first, j2_last;    /* local columns the half-step-lattice sweep writes       */
} slb_step_geom;

/* One whole loop iteration in one launch: step_on_grid followed by
 * step_on_half_grid (reference: src/boltzmann_solver.c:204-217), with the main
 * sweep's output handed to the half-step sweep inside the kernel instead of
 * through memory.  Results are bit-identical to the two separate calls.
 * cosines[0..1] belong to the main sweep, [2..3] to the half-step sweep.
 * Every column the half-step sweep reads at m+-1 must either be written by the
 * main sweep of this call (j1 range) or be a column no sweep ever writes
 * (then its present contents are used, as in the two-call form).  A
 * single-GPU step is {PADDED_MSIZE, 0, 1, M+1, 1, M}.  Returns like slb_sweep. */
int slb_step_fused(const slb_step_geom *geom,
                   const float *a0, const float *a_current, const float *b_current,
                   const float *a_current_hs, const float *b_current_hs,
                   float *a_next, float *b_next, float *a_next_hs, float *b_next_hs,
                   const float cosines[4]);

/* Geometry of a double step: the columns each of the four sweeps produces
 * (0 = main sweep at t, 1 = half-step sweep at t, 2 = main sweep at t+dt,
 * 3 = half-step sweep at t+dt).  A single-GPU double step is
 * {PADDED_MSIZE, 0, {1,1,1,1}, {M+1,M,M+1,M}}. */
typedef struct slb_step2_geom {
  int pitch, m_offset;
  int j_first[4], j_last[4];
} slb_step2_geom;

/* TWO loop iterations in one launch (reference: two passes of the loop body
 * src/boltzmann_solver.c:204-217 + :252-253).  The lattices at t+dt and
 * t+3dt/2 are handed from sweep to sweep inside the kernel and never stored;
 * a_out/b_out receive the main lattice at t+2dt, a_out_hs/b_out_hs the
 * half-step lattice at t+5dt/2, bit-identical to four separate sweeps.
 * a_next/b_next and a_next_hs/b_next_hs are only READ, at the cells no sweep
 * writes (row N, boundary columns), where the reference's second iteration
 * would find them.  The out arrays must not alias the inputs, and their
 * never-written cells must already equal those of a_current/b_current
 * (resp. the *_current_hs arrays): the caller keeps such spare arrays and
 * exchanges them with the current ones after the call.  cosines[0..3] are the
 * first iteration's, [4..7] the second's.  Returns like slb_sweep. */
int slb_step_fused2(const slb_step2_geom *geom, const float *a0,
                    const float *a_current, const float *b_current,
                    const float *a_current_hs, const float *b_current_hs,
                    const float *a_next, const float *b_next,
                    const float *a_next_hs, const float *b_next_hs,
                    float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                    const float cosines[8]);

/* slb_step_fused2 that also hands out what av() needs of BOTH new main lattices:
 * rows_first and rows_second (device, [3][pitch] floats each, both or neither)
 * receive rows 0 and 1 of a and row 1 of b -- as buffer rows 0, 1, 2 -- of the
 * main lattice after the first and after the second of the two steps, at the
 * columns those sweeps produce.  The lattice after the first step is stored
 * nowhere else.  slb_av_sums_exact(slab, rows, rows + pitch, tot) then forms
 * that step's sums from such a buffer (it looks for b's row 1 at b + pitch,
 * which is buffer row 2). */
int slb_step_fused2_rows(const slb_step2_geom *geom, const float *a0,
                         const float *a_current, const float *b_current,
                         const float *a_current_hs, const float *b_current_hs,
                         const float *a_next, const float *b_next,
                         const float *a_next_hs, const float *b_next_hs,
                         float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                         const float cosines[8], float *rows_first, float *rows_second);

/* The equilibrium coefficients in separable form.  a0[n][m] is the product of a factor of the
 * row and a factor of the column (reference: src/boltzmann_c_solver.c:117-122: a float
 * coefficient of n times a long-double Gaussian of phi_y(m), narrowed to float), so a kernel
 * that walks rows for fixed columns can form it -- (float)(a0_row[n] * a0_col[j]), one
 * double-precision product -- instead of reading it: 32 instead of 36 bytes per cell and
 * double step.  Whether the product reproduces the stored float in EVERY cell depends on the
 * factors' last bits, so it is checked, never assumed: slb_a0_factors_check() counts, for rows
 * row_first .. row_first+nrows-1 and columns 0 .. ncols-1, the cells in which the two differ
 * (bad[n] += count; device ints, cleared by the caller).  A caller enables the factored form
 * only when every count is zero (host/slb_solver.c also tries neighbouring doubles as the row
 * factor of a row that fails), and the lattices are then bit-identical by construction. */
int slb_a0_factors_check(const float *a0, int pitch, int row_first, int nrows, int ncols,
                         const double *a0_row, const double *a0_col, int *bad);

/* slb_step_fused2_rows (rows_first/rows_second may both be NULL) with a0 formed from checked
 * factors: a0_row has N+1 doubles, a0_col `pitch` doubles (16-byte aligned, local columns).
 * The stored a0 is still passed: launches too narrow to fill the device go to the general
 * kernel, which reads it. */
int slb_step_fused2_factored(const slb_step2_geom *geom, const float *a0, const double *a0_row, const double *a0_col,
                             const float *a_current, const float *b_current,
                             const float *a_current_hs, const float *b_current_hs,
                             const float *a_next, const float *b_next,
                             const float *a_next_hs, const float *b_next_hs,
                             float *a_out, float *b_out, float *a_out_hs, float *b_out_hs,
                             const float cosines[8], float *rows_first, float *rows_second);

/* Self-test of the shared-reciprocal division the tuned kernels use for the two quotients of
 * a point (same denominator): `count` pseudo-random triples (two numerators over every bit
 * pattern -- denormals, zeros, infinities, NaN included -- and a denominator between 0.5 and
 * 2^100, the range the kernels use it in) are divided both ways on the GPU; *differ (device, cleared by the caller) receives the
 * number of quotients whose bits differ from the compiler's IEEE division, first_bad (device,
 * 4 floats) the first such numerator, denominator and the two results. */
int slb_selftest_division(unsigned long long count, unsigned seed, unsigned long long *differ, float *first_bad);

/* Test hook, host arithmetic only (no GPU call; no counterpart in the reference): the launch
 * schedule a multi-step kernel would get for rows 0..N-1 and columns col_first..col_last on a
 * device that holds `slots` of its workgroups at once (0: unknown), and who forms what under it.
 * kind 2: the double step's schedule; 3: the S-step kernels' (steps = 2 or 3) with the two edge
 * strip groups first; 4: the same without; 5: as 3, with whole strips also below one round of workgroups
 * (tuning variant 3).  block: threads per workgroup; n_chunk > 0: every
 * strip group cut into chunks of that many rows.  Returns the number of workgroups (< 0: bad
 * arguments, a strip group out of range, cover too short); info[0..3] = strip groups, phases,
 * columns per strip group, first column of strip group 0; cover[g * N + n] = how many workgroups
 * form row n of strip group g -- a correct schedule leaves 1 everywhere. */
int slb_debug_schedule(int kind, int steps, int N, int col_first, int col_last, int block, int n_chunk,
                       int slots, int *info, int *cover, int cover_len);

/* Geometry of S = `steps` loop iterations in one launch (1 <= S <= 4): the
 * columns each of the 2S sweeps produces, in execution order (main sweep at t,
 * half-step sweep at t, main sweep at t+dt, ...). */
typedef struct slb_multi_geom {
  int pitch, m_offset, steps;
  int j_first[8], j_last[8];
} slb_multi_geom;

/* S loop iterations in one launch, the general form of slb_step_fused (S=1) and
 * slb_step_fused2 (S=2); bit-identical to 2S separate sweeps.
 *   in[4]     = a_current, b_current, a_current_hs, b_current_hs
 *   out[4]    = main lattice after S iterations (a, b), half-step lattice (a, b);
 *               must not alias `in`
 *   pers_a/pers_b[q], q = 0..2S-2: for sweep q, the arrays that hold -- at the
 *               cells no sweep writes -- what the reference's ping-pong buffers
 *               hold where sweep q's output would go (a/b[next], a/b[next_hs],
 *               a/b[current], a/b[current_hs], a/b[next], ...); entry 2S-2 is out[0..1]
 *   cosines[4S]: slb_step_cosines() of the S loop times, concatenated.
 * Only the last two sweeps' outputs are stored.  Returns like slb_sweep. */
int slb_step_multi(const slb_multi_geom *geom, const float *a0, const float *const in[4],
                   const float *const pers_a[], const float *const pers_b[], float *const out[4],
                   const float cosines[]);

/* The same for S = 1..3, also writing the rows av() reads -- a rows 0 and 1, b row 1 -- of the
 * main lattice after EACH of the S iterations into rows[i] (i = 0..S-1; device, [3][pitch]
 * floats each, a row 0 first), at the columns the launch forms: the lattices in between are
 * stored nowhere else, so this is what lets the averaging period (reference:
 * src/boltzmann_c_solver.c:188-191) keep S iterations per launch. */
int slb_step_multi_rows(const slb_multi_geom *geom, const float *a0, const float *const in[4],
                        const float *const pers_a[], const float *const pers_b[], float *const out[4],
                        const float cosines[], float *const rows[]);

/* Gather local columns col..col+ncols-1 of rows 0..nrows-1 of arrays a and b
 * into buf (a's columns first, each column contiguous: buf[(q*ncols + c)*nrows + r],
 * q = 0 for a, 1 for b); and the inverse scatter.  buf holds 2*ncols*nrows floats. */
int slb_halo_pack  (const float *a, const float *b, int pitch, int nrows, int col, int ncols, float *buf);
int slb_halo_unpack(float *a, float *b, int pitch, int nrows, int col, int ncols, const float *buf);

/* The same for up to four (array pair, column run, buffer) segments in ONE launch:
 * what a slab sends to (or receives from) both neighbours after a step, main and
 * half-step lattice together.  Each segment's buffer is laid out as above. */
typedef struct slb_halo_seg {
  float *a, *b;     /* the two arrays, [nrows][pitch] */
  int    col, ncols;
  float *buf;       /* 2*ncols*nrows floats */
} slb_halo_seg;
int slb_halo_pack_segs  (const slb_halo_seg *segs, int nsegs, int pitch, int nrows);
int slb_halo_unpack_segs(const slb_halo_seg *segs, int nsegs, int pitch, int nrows);

/* First stage of av(): the three sums over local columns j_first..j_last
 * (b[1][m]*dPhi, a[0][m]*phi_y(m)*dPhi, a[1][m]*dPhi), terms formed in float as
 * the reference does, accumulated in double by a wave64 shuffle tree.
 * sums[0..2] are overwritten. */
int slb_av_partial(const slb_slab *slab, const float *a, const float *b, double *sums);

/* Second stage of av(): narrow sums[0..2] to float and apply the reference's
 * update of av_data[0..5] at loop time t. */
int slb_av_update(const double *sums, float *av_data, float t);

/* av() in two calls, for hosts that overlap the sums with later steps: the three
 * exact-order sums of columns slab->j_first..j_last into tot3[0..2] (device
 * floats; nothing shared between calls, so several may run on different streams
 * at once), and the update of av_data[0..5] at loop time t from such sums. */
int slb_av_sums_exact(const slb_slab *slab, const float *a, const float *b, float *tot3);
/* the same for n <= 8 lattices in one launch (a[k], b[k] -> tot3[k]) */
int slb_av_sums_exact_n(const slb_slab *slab, int n, const float *const a[], const float *const b[],
                        float *const tot3[]);
/* the same with the running sums starting from init3[k][0..2] instead of zero (init3 or
 * init3[k] NULL: zero; init3[k] == tot3[k] is allowed): the sums of a lattice cut into
 * column ranges are the chain range 1 -> range 2 -> ..., each call starting from the
 * totals of the one before -- how phi_y slabs reproduce the reference's single running
 * sum over all columns (reference: src/boltzmann_c_solver.c:413-437) bit for bit */
int slb_av_sums_exact_from_n(const slb_slab *slab, int n, const float *const a[], const float *const b[],
                             const float *const init3[], float *const tot3[]);
int slb_av_update_exact(const float *tot3, float *av_data, float t);

/* How av() forms its three sums:
 *   SLB_AV_EXACT (default) -- the terms are added in float in the reference's
 *       order, bit-identical to the reference CPU av(), but by a whole
 *       workgroup per sum: inside one binade of the running sum every addition
 *       rounds to one fixed grid, so 4096 terms at a time become an integer
 *       prefix sum; ties, binade changes and special values take a real float
 *       addition (see slb_seq_sum);
 *   SLB_AV_CHAIN -- the same bits from one lane per sum adding term after term
 *       (a chain of M dependent additions; kept as a cross-check);
 *   SLB_AV_TREE -- slb_av_partial + slb_av_update: wave64 shuffle tree in
 *       double; differs from the CPU's sequential float sum by that sum's own
 *       rounding noise (about 1e-4 relative at M = 2^20 for a sum that nearly
 *       cancels, 1e-6 for one that does not).  At large M this mode therefore does
 *       NOT keep the averaged drift velocity within 1e-6 relative of the CPU
 *       solver's; only SLB_AV_EXACT and SLB_AV_CHAIN do.  Opt-in, never the default. */
enum { SLB_AV_EXACT = 0, SLB_AV_TREE = 1, SLB_AV_CHAIN = 2 };
void slb_set_av_mode(int mode);

/* display=8 frame values, the inner loops of the reference's print_2d_data
 * (src/boltzmann_solver.c:495-504) without the text formatting:
 *   out[p*ncols + c] = max(0, sum_{n=0}^{nrows-1} a[n][m] cos(n phi_x[p]) + b[n][m] sin(n phi_x[p])),
 *   m = col_first + c, for p = 0..nphi-1 and c = 0..ncols-1.
 * cs (device, 16-byte aligned) holds the trigonometric factors as the host's
 * libm gives them: cs[2*(p*nrows + n)] = cos((double)(n*phi_x[p])), the next
 * double the sine, the product n*phi_x[p] formed in float like the reference.
 * The running value is a float; each term is formed and added in double and
 * rounded back after every harmonic, which reproduces the reference's value
 * bit for bit.  a, b: device arrays [nrows][pitch]; out: device, nphi*ncols floats.
 * b == NULL leaves the sine terms out: the sum display=3 forms over the
 * equilibrium coefficients a0 (reference: src/boltzmann_solver.c:322). */
int slb_frame_values(const float *a, const float *b, int pitch, int nrows, int col_first, int ncols,
                     const double *cs, int nphi, float *out);

/* out[0] = the float the loop `acc = 0; for (i = 0; i < n; i++) acc = acc + x[i];`
 * leaves in acc (IEEE single, round to nearest even, denormals kept), bit for
 * bit, for any contents of x (device, n floats) -- evaluated by one workgroup
 * in parallel, the way SLB_AV_EXACT forms the sums of av(). */
int slb_seq_sum(const float *x, int n, float *out);
/* The same, and how it went: stats[0] = parallel passes taken, stats[1] = runs of
 * terms that one thread added one by one (device, 2 ints; may be NULL). */
int slb_seq_sum_stats(const float *x, int n, float *out, int *stats);

/* av() for callers that cannot allocate device memory: code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  av()
 * keeps its reduction workspace (block partials of the tree sums, the totals handed from
 * the sums to the running-average update) in memory it allocates on first use;
 * slb_av_static() runs the same kernels in the same av mode with that workspace in a
 * statically allocated device array (24 608 bytes on every GPU the library runs on, held
 * until the process ends) and calls no hipMalloc, hipFree or other allocating routine.
 * Same arguments as av() without `blocks`, same effect on av_data[0..5], bit for bit.
 * The workspace does not grow with the lattice (the tree sums use at most 1024 blocks of
 * partials whose threads stride over the columns, the exact sums none), so every g-grid av()
 * takes is taken here too; the call refuses only what av() cannot do either (load_data() not
 * called yet, a null array) -- it prints the reason and returns -1 without launching.
 * Returns 0, -1 for an argument error, or a hipError_t value.  All calls on one GPU share
 * that GPU's array: calls that overlap there (different streams, different threads) race;
 * calls on one stream do not.  slb_static_init() looks the array's device address up for the
 * current device (also done by the first slb_av_static() there); call it once per device
 * at set-up if the first slb_av_static() would otherwise fall inside a graph capture, where
 * the whole call then records as kernel nodes only. */
int slb_static_init(void);
int slb_av_static(float *a, float *b, float *av_data, float t);

/* Launch stream (a hipStream_t passed as void*).  slb_set_stream() sets the default of the
 * whole process (the reference's boundary has no stream argument; default: the null stream).
 * slb_push_stream() makes a stream the launch stream of the CALLING THREAD until the matching
 * slb_pop_stream() (up to 8 nested): a host that works with several streams brackets each
 * call with the stream it is meant for instead of changing process-wide state.
 * slb_get_stream(): where the calling thread's next launch would go. */
void  slb_set_stream(void *hip_stream);
void *slb_get_stream(void);
void  slb_push_stream(void *hip_stream);
void  slb_pop_stream(void);

/* Kernel selection and launch shape, for tuning; 0 = let the library choose.
 * variant, slb_sweep: 1 = one column per lane (any pitch), 2 = four columns per
 *   lane with neighbour columns re-read through L1, 3 = neighbour columns from
 *   the adjacent lanes, 4 = software-pipelined form of 2, 5 = 4 with nontemporal
 *   accesses for data touched once (the default where rows are 16-byte aligned).
 *   9 = 4 with nontemporal stores only.
 * variant, slb_step_fused: 6 = plain accesses (default), 7 = nontemporal, 8 =
 *   nontemporal stores only.
 * variant, slb_step_fused2 (default: two halo lanes per side of a wave, nontemporal
 *   stores, registers for 3 waves per SIMD, whole column strips first from one round
 *   of workgroups up): 20 = one halo lane and plain stores, 21 / 25 = 20 with a
 *   register budget of 2 / 4 waves per SIMD, 27 / 23 = one / four halo lanes, 29 / 30
 *   = two / four halo lanes with plain stores, 24 = nontemporal loads too, 26 = every
 *   strip cut into chunks of rows, 34 = every strip cut below four rounds of workgroups,
 *   38 = the four-column kernel also for launches too narrow to fill the device (which
 *   otherwise go to the general kernel with one column per lane), 39 = the schedule taken
 *   when the device's occupancy cannot be queried (every strip cut into chunks of rows).
 *   60 = the point update with the row-dependent factors taken out of the general rows and dt*a0
 *   handed from stage to stage, 61 = 60 with the two quotients of a point sharing one reciprocal.
 *   slb_step_fused2_factored loads a0 like slb_step_fused2_rows unless asked otherwise (forming
 *   it measured slower on MI355X): 50 = a0 formed from the factors once per row and handed from
 *   stage to stage, 51 = every stage forms the a0 values of its own row.
 *   64 = the compiler's own point update (default: two columns at a time with one reciprocal for
 *   the two quotients of a point, where the run's denominators lie between 0.5 and 2^100).
 *   slb_step_multi (default at three steps per launch: the LDS-ring kernel with branch-free steady
 *   passes): 1 = the register-history kernel at any S; 2 = the default kernel with one round of whole
 *   column strips fewer, the strips left over in chunks of n_chunk rows (measured slower); 3 = on a lattice of less
 *   than one round of workgroups whole column strips all the same (edge strip groups first; measured no faster); 32 / 33 / 34 = that kernel with two columns
 *   per lane instead of four, with a register budget of 2 / 3 / 4 waves per SIMD; 35 / 36 / 37 = one
 *   column per lane, budget of 2 / 4 / 8 waves per SIMD; 70..75 = the scheduled kernel with its
 *   stage outputs in registers, 80..84 = in an LDS ring per wave (S = 2, 3); 90 / 91 = the ring
 *   kernel with branch-free steady passes at S = 2 (budget of 2 / 3 waves per SIMD) and S = 3.
 *   slb_seq_sum: 41..47 = other workgroup shapes.
 * n_chunk = rows per workgroup in n (0: by lattice width), block = threads (0: 256). */
void slb_set_tuning(int variant, int n_chunk, int block);

/* Library / device identification for logs: returns a static string. */
const char *slb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SLB_ABI_H */
