// chain_tasks.h -- the persistent panel chain's ticket -> task map, shared by the kernel (chol.hip, chain_kernel), the
// host side that sizes the launch (launch_chain) and the C ABI's test hook (tgp_chain_task: tests/test_chain_tasks.py
// enumerates every ticket of a launch on the CPU and checks that each task exists exactly once and waits for EARLIER
// tickets only -- what makes one-task-per-workgroup deadlock-free without co-residency).  CHAIN_HD is defined by the
// includer (__host__ __device__ in HIP code, empty in a host-only translation unit).
//
// Round 6: the kernel no longer decodes its ticket.  launch_chain builds the launch's task list ONCE per shape on the host
// (chain_build below: this header's order, with the K-BATCHED updates of ChainPolicy folded in), keeps it on the device as a
// table of packed 64-bit words and a workgroup reads tasks[ticket].  The walk below is host code now (it cost every
// workgroup 1-3 us of scalar work at its start); the device needs the packing and CHAIN_CRIT_PARTS only.
#pragma once
#include <cstdint>
#include <vector>

struct ChainTask {
  int kind;  // 0 solve(i, c) | 1 diag(c) | 2 update(i, c, k) | 3 update of the diagonal tile (c, c) from column k |
             // 4 one of CHAIN_CRIT_PARTS parts (`part`) of update(i, c, k) | 5 xsolve(c): the streamed solve of tile (c, c-1) |
             // 6 update(i, c, [k0, k]): ONE product over block columns k0 .. k (K = 128 (k - k0 + 1)), ChainPolicy |
             // 7 fsolve(c): z_c = L_cc^-1 y_c | 8 fupdate(c, g): y_i -= X_ic z_c for the row tiles i > c of group g = `i` | -1 none
  int i, c, k, part;
  int k0;    // kind 6: first block column of the batch (else = k)
};
// one table word: kind 4 bits | part 4 | c 8 | k 8 | k0 8 | i 16
CHAIN_HD inline uint64_t chain_pack(const ChainTask& t) {
  return uint64_t(t.kind & 15) | (uint64_t(t.part & 15) << 4) | (uint64_t(t.c & 255) << 8) | (uint64_t(t.k & 255) << 16) |
         (uint64_t(t.k0 & 255) << 24) | (uint64_t(t.i & 65535) << 32);
}
CHAIN_HD inline ChainTask chain_unpack(uint64_t w) {
  ChainTask t;
  t.kind = int(w & 15); t.part = int((w >> 4) & 15); t.c = int((w >> 8) & 255); t.k = int((w >> 16) & 255);
  t.k0 = int((w >> 24) & 255); t.i = int((w >> 32) & 65535);
  return t;
}
constexpr int CHAIN_CRIT_PARTS = 8;  // workgroups that share the update of tile (k+2, k+1) from column k

// TICKET ORDER (round 5).  With DG(j) = [xsolve(j), diag(j)] (block j is factored by this launch: cb < j < ce),
// Solves(k) = solve(i, k) for the rows below (i from k+2 while xsolve(k+1) owns tile (k+1, k)), and the updates FROM column k
//   Crit(k)   update(k+2, k+1, k)   in CHAIN_CRIT_PARTS parts -- the tile xsolve(k+2) waits for
//   D(k)      update(k+2, k+2, k)   the diagonal tile diag(k+2) factors
//   U1r(k)    update(i,   k+1, k), i >= k+3    the rest of the column the next step solves
//   E(k)      update(k+3, k+2, k), update(k+3, k+3, k)    what Crit(k+1) and D(k+1) build on
//   Bulk(k)   every other update(i, c, k), c >= k+2
// a launch hands out
//   [diag(cb) if cb > 0]  DG(cb+1)  Solves(cb) Crit(cb) D(cb)
//   then for k = cb, cb+1, ...:   U1r(k) E(k) | DG(cb+2) if k == cb | Solves(k+1) Crit(k+1) D(k+1) | DG(k+3) | Bulk(k)
// when the launch covers the WHOLE REST of the matrix (nblk == R: the one-launch tail, N <= 4 096) -- chain-bound by design.
// A panel's launch (8 block columns, up to 128 row tiles) keeps the solves in their own step:
//   ... for k = cb, cb+1, ...:   Solves(k) Crit(k) U1r(k) D(k) | DG(k+2) | Bulk(k) with E(k)
// -- there only the two diagonal tasks move in front of the bulk: a hundred solves spinning for L_{k+1,k+1} beside a big
// trailing update cost more than the lane gains (one box, profiles/r05_r: N = 4 096 1.378 lane / 1.393 DG-only / 1.422 round 4;
// c2 25.66 / 25.40 / 25.63).
// -- the DIAGONAL LANE of the next step (its solves, the two tiles the block after next needs, that block's xsolve and
// potf2) sits IN FRONT of the bulk of this step.  Until round 4 a step was [DG | Solves | all updates]: the diagonal tasks of
// step k+1 drew their tickets behind the (R-k)^2 / 2 bulk updates of step k -- ~460 tasks of 22 us on 256 compute units at
// N = 4 096 -- although they need none of them; the stamped timeline showed xsolve / diag starting 15-37 us late and the
// first ~11 blocks of a 32-block launch at 45-65 us per block instead of 36.  Everything a task of the lane waits for has
// an earlier ticket in this order too (tests/test_chain_tasks.py checks every wait of every task of a launch).
// FORWARD SUBSTITUTION AS TASKS (round 6).  The fused log_probability needs z = L^-1 y; until round 5
// step c of it -- z_c = L_cc^-1 y_c, y_i -= X_ic z_c below -- was a pair of small launches on the solve stream behind a
// one-wave POLL kernel that watched block column c of the running chain launch (128 pollers + 255 launches per evaluation
// at N = 16 384).  Now it is part of the launch:
//   F(c) = fsolve(c), fupdate(c, g) for every group g of `gs` row tiles that holds rows below c + 1
// handed out at the END of step c (behind its bulk: nobody waits for z but the next F).
//   fsolve(c)     y_c -= X_{c,c-1} z_{c-1} (the ONE tile between z_{c-1} and z_c: the dependent chain of the substitution is
//                 a tile product and a 128 x 128 solve per block, ~10 us against the factorisation's 36), then z_c = L_cc^-1 y_c;
//                 waits for L_cc, z_{c-1} and for row c to carry the columns before c - 1;
//   fupdate(c, g) y_i -= X_ic z_c for the rows i > c + 1 of group g; waits for z_c, for block column c to be final and for
//                 fupdate(c-1, g) -- the updates of a row arrive in COLUMN ORDER whatever runs where: the sums are fixed.
// (Row c + 1 of the LAST block column of a panel belongs to the next panel's launch: its fsolve(0) applies that tile.)
struct ChainLaunch {
  int R, nblk, cb, ce;
  int lane;  // 1: the next step's whole diagonal lane in front of the bulk; 0: its two diagonal tasks only
  int fwd;   // 1: the forward substitution of the launch's block columns rides along as tasks (kinds 7, 8)
  int gs;    // row tiles per fupdate task
};
constexpr int CHAIN_FWD_GROUP = 16;
CHAIN_HD inline ChainLaunch chain_launch(int R, int nblk, int cb, int ce, int fwd = 0, int gs = CHAIN_FWD_GROUP) {
  ChainLaunch q = {R, nblk, cb, ce, nblk == R ? 1 : 0, fwd, gs};
  return q;
}
// groups of row tiles with a row below k + 1: (k+2) / gs .. (R-1) / gs
CHAIN_HD inline int chain_fwd_g0(const ChainLaunch& q, int k) { return (k + 2) / q.gs; }
CHAIN_HD inline int chain_n_F(const ChainLaunch& q, int k) {
  if (!q.fwd || k < q.cb || k >= q.ce) return 0;
  return 1 + (k + 2 <= q.R - 1 ? (q.R - 1) / q.gs - chain_fwd_g0(q, k) + 1 : 0);
}
CHAIN_HD inline bool chain_factored(const ChainLaunch& q, int j) { return j > q.cb && j < q.ce; }  // DG(j) exists
CHAIN_HD inline int chain_solve_r0(const ChainLaunch& q, int k) { return chain_factored(q, k + 1) ? k + 2 : k + 1; }
CHAIN_HD inline int chain_n_solves(const ChainLaunch& q, int k) {
  const int n = q.R - chain_solve_r0(q, k);
  return n > 0 ? n : 0;
}
// update(i, c, k) exists for k+1 <= c <= nblk-1, c <= i <= R-1, except tile (k+1, k+1) (diag(k+1)'s own fold)
CHAIN_HD inline bool chain_has_update(const ChainLaunch& q, int i, int c, int k) {
  return c >= k + 1 && c <= q.nblk - 1 && i >= c && i <= q.R - 1 && !(i == k + 1 && c == k + 1);
}
CHAIN_HD inline int chain_n_crit(const ChainLaunch& q, int k) { return chain_has_update(q, k + 2, k + 1, k) ? CHAIN_CRIT_PARTS : 0; }
CHAIN_HD inline int chain_n_D(const ChainLaunch& q, int k) { return chain_has_update(q, k + 2, k + 2, k) ? 1 : 0; }
CHAIN_HD inline int chain_n_U1r(const ChainLaunch& q, int k) {
  if (k + 1 > q.nblk - 1) return 0;
  const int n = q.R - (k + 3);
  return n > 0 ? n : 0;
}
CHAIN_HD inline int chain_n_E(const ChainLaunch& q, int k) {
  return (chain_has_update(q, k + 3, k + 2, k) ? 1 : 0) + (chain_has_update(q, k + 3, k + 3, k) ? 1 : 0);
}
CHAIN_HD inline int chain_n_bulk(const ChainLaunch& q, int k) {
  const int m = q.nblk - 1 - (k + 2) + 1;  // columns k+2 .. nblk-1
  if (m <= 0) return 0;
  const int n = m * q.R - ((k + 2 + q.nblk - 1) * m) / 2;  // sum of (R - c): (first + last) * m is even or m is
  return n - chain_n_D(q, k) - (q.lane ? chain_n_E(q, k) : 0);
}

// one walk over the order above: `t` < 0 counts the tickets (returned), `t` >= 0 decodes that ticket into *out;
// `all` != NULL (host): every task of the launch is appended in ticket order (one walk, not one per ticket)
typedef std::vector<ChainTask> ChainTaskList;
inline int64_t chain_walk(const ChainLaunch& q, int64_t t, ChainTask* out, ChainTaskList* all = nullptr) {
  int64_t seen = 0;
  ChainTask task = {-1, 0, 0, 0, 0, 0};
#define CHAIN_EMIT_ALL(cnt_, body)                        \
  if (all != nullptr) {                                   \
    for (int u = 0; u < int(cnt_); ++u) {                 \
      task = ChainTask{-1, 0, 0, 0, 0, 0};                \
      body;                                               \
      task.k0 = task.k;                                   \
      all->push_back(task);                               \
    }                                                     \
  }
#define CHAIN_GROUP(count, body)                          \
  do {                                                    \
    const int64_t cnt_ = (count);                         \
    CHAIN_EMIT_ALL(cnt_, body)                            \
    if (t >= 0 && t < seen + cnt_) {                      \
      const int u = int(t - seen);                        \
      (void)u;                                            \
      body;                                               \
      task.k0 = task.k;                                   \
      *out = task;                                        \
      return seen + cnt_;                                 \
    }                                                     \
    seen += cnt_;                                         \
  } while (0)
  auto dg = [&](int j, int u) { task.kind = u == 0 ? 5 : 1; task.c = j; };
  auto solve = [&](int k, int u) { task.kind = 0; task.i = chain_solve_r0(q, k) + u; task.c = k; };
  auto crit = [&](int k, int u) { task.kind = 4; task.i = k + 2; task.c = k + 1; task.k = k; task.part = u; };
  auto dtile = [&](int k) { task.kind = 3; task.i = task.c = k + 2; task.k = k; };
  auto u1r = [&](int k, int u) { task.kind = 2; task.i = k + 3 + u; task.c = k + 1; task.k = k; };
  auto etile = [&](int k, int u) {
    const bool first = chain_has_update(q, k + 3, k + 2, k);
    task.k = k;
    task.i = k + 3;
    if (first && u == 0) { task.kind = 2; task.c = k + 2; }
    else { task.kind = 3; task.c = k + 3; }
  };
  auto bulk = [&](int k, int u) {  // column by column; D(k) and E(k) are the FIRST rows of columns k+2 and k+3: skipped
    for (int c = k + 2; c <= q.nblk - 1; ++c) {
      int skip = 0;
      if (c == k + 2) skip = (chain_has_update(q, k + 2, k + 2, k) ? 1 : 0) + ((q.lane && chain_has_update(q, k + 3, k + 2, k)) ? 1 : 0);
      else if (c == k + 3) skip = (q.lane && chain_has_update(q, k + 3, k + 3, k)) ? 1 : 0;
      const int cnt = q.R - c - skip;
      if (u < cnt) {
        task.i = c + skip + u;
        task.c = c;
        task.k = k;
        task.kind = task.i == c ? 3 : 2;
        return;
      }
      u -= cnt;
    }
  };
  auto fwd = [&](int k, int u) {
    task.c = k;
    if (u == 0) { task.kind = 7; }
    else { task.kind = 8; task.i = chain_fwd_g0(q, k) + u - 1; }
  };
  if (q.cb > 0) CHAIN_GROUP(1, { task.kind = 1; task.c = q.cb; });
  const int k0 = q.cb;
  if (chain_factored(q, k0 + 1)) CHAIN_GROUP(2, dg(k0 + 1, u));
  if (!q.lane) {
    for (int k = k0; k < q.ce; ++k) {
      CHAIN_GROUP(chain_n_solves(q, k), solve(k, u));
      CHAIN_GROUP(chain_n_crit(q, k), crit(k, u));
      CHAIN_GROUP(chain_n_U1r(q, k), u1r(k, u));
      CHAIN_GROUP(chain_n_D(q, k), dtile(k));
      if (chain_factored(q, k + 2)) CHAIN_GROUP(2, dg(k + 2, u));
      CHAIN_GROUP(chain_n_bulk(q, k), bulk(k, u));
      CHAIN_GROUP(chain_n_F(q, k), fwd(k, u));
    }
    return seen;
  }
  CHAIN_GROUP(chain_n_solves(q, k0), solve(k0, u));
  CHAIN_GROUP(chain_n_crit(q, k0), crit(k0, u));
  CHAIN_GROUP(chain_n_D(q, k0), dtile(k0));
  for (int k = k0; k < q.ce; ++k) {
    CHAIN_GROUP(chain_n_U1r(q, k), u1r(k, u));
    CHAIN_GROUP(chain_n_E(q, k), etile(k, u));
    if (k == k0 && chain_factored(q, k + 2)) CHAIN_GROUP(2, dg(k + 2, u));
    if (k + 1 < q.ce) {  // column k+1 is solved by this launch: its diagonal lane
      CHAIN_GROUP(chain_n_solves(q, k + 1), solve(k + 1, u));
      CHAIN_GROUP(chain_n_crit(q, k + 1), crit(k + 1, u));
      CHAIN_GROUP(chain_n_D(q, k + 1), dtile(k + 1));
    }
    if (chain_factored(q, k + 3)) CHAIN_GROUP(2, dg(k + 3, u));
    CHAIN_GROUP(chain_n_bulk(q, k), bulk(k, u));
    CHAIN_GROUP(chain_n_F(q, k), fwd(k, u));
  }
#undef CHAIN_GROUP
#undef CHAIN_EMIT_ALL
  return seen;
}

// tickets of a launch over block columns [cb, ce) of a panel with R row tiles and nblk block columns.  A continuation
// launch (cb > 0) starts with diag(cb): tile (cb, cb-1) is final since the launch before; a panel's very first block
// (cb == 0) is factored in front of the launch.
CHAIN_HD inline int64_t chain_task_count(int R, int nblk, int cb, int ce) {
  const ChainLaunch q = chain_launch(R, nblk, cb, ce);
  return chain_walk(q, -1, nullptr);
}

CHAIN_HD inline ChainTask chain_decode_ticket(int t, int R, int nblk, int cb, int ce) {
  const ChainLaunch q = chain_launch(R, nblk, cb, ce);
  ChainTask task = {-1, 0, 0, 0, 0, 0};
  chain_walk(q, t, &task);
  return task;
}

// ---- K-BATCHED updates (round 6) ------------------------------------------------------------------------------------
// A right-looking update task reads and writes its 128 x 128 tile for ONE 128-deep product: 256 KB of tile traffic per
// 4.2 MFLOP.  A launch over many block columns (the one-launch tail) is bound by exactly that -- N = 8 192 as one launch ran
// at 27 TFLOP/s (profiles/r06_a) -- and the tiles far from the diagonal are in no hurry: tile (i, c) is needed when the
// chain reaches column c, and its row when it reaches row i.  So the updates of a tile are BATCHED along k: tile (i, c)
// has boundaries at the block columns b with (b - cb + (i + c)) % batch == 0 -- staggered over the tiles so that every
// step carries the same share of batch tasks -- and ONE task applies the columns between two boundaries,
// update(i, c, [k0, k1)), K = 128 (k1 - k0): the tile is read and written once per batch (and the 4x4x4 MFMA form's
// prologue / publish are paid once: chain_update_fast).  A batch exists when
//   i >= c + rowlag      the tile is off the diagonal lane (tiles near the diagonal feed xsolve / diag within a step or two),
//   k1 + lag <= c        its last column is solved at least lag + 1 steps before column c is,
//   k1 <= ce, k1 - k0 >= 2,
//   R - k1 >= minrows    the launch is still THROUGHPUT-bound when the batch is handed out: (R - k)^2 / 2 update tasks per
//                        step fill 256 compute units for longer than a step of the diagonal chain while more than ~30 row
//                        tiles are left; behind that the chain sets the pace, a 70-us batch task only delays the tiles the
//                        diagonal lane needs next (N = 4 096 as one launch: 1.35 -> 1.81 ms with batches everywhere,
//                        profiles/r06_b), and every update stays the single task it was
// and takes the ticket update(i, c, k1 - 1) had (everything it waits for -- X_{i,k1-1}, X_{c,k1-1}, the tile's previous batch
// or update -- has an earlier ticket there: tests/test_chain_tasks.py); the updates it absorbs disappear.  Every other
// update stays the single task it was.  The order of summation inside a tile changes with the policy (deterministic for
// a given policy; the results of different policies agree to rounding).
struct ChainPolicy {
  int batch;    // block columns per batch (<= 1: off)
  int lag;      // >= 1
  int rowlag;   // >= 2
  int minrows;  // row tiles that must be left behind the batch's last column (0: no such condition)
};
CHAIN_HD inline bool chain_batch_of(const ChainLaunch& q, const ChainPolicy& p, int i, int c, int k, int* k0, int* k1) {
  if (p.batch <= 1 || i < c + p.rowlag) return false;
  const int r = (k + 1 - q.cb + i + c) % p.batch;
  const int e = r == 0 ? k + 1 : k + 1 + (p.batch - r);  // first boundary behind column k
  const int b = e - p.batch > q.cb ? e - p.batch : q.cb;
  if (e + p.lag > c || e > q.ce || e - b < 2 || q.R - e < p.minrows) return false;
  *k0 = b;
  *k1 = e;
  return true;
}

// the launch's task list in ticket order, batches folded in (host)
inline std::vector<ChainTask> chain_build(int R, int nblk, int cb, int ce, const ChainPolicy& p, int fwd = 0,
                                          int gs = CHAIN_FWD_GROUP) {
  const ChainLaunch q = chain_launch(R, nblk, cb, ce, fwd, gs);
  std::vector<ChainTask> all, out;
  chain_walk(q, -1, nullptr, &all);
  out.reserve(all.size());
  for (ChainTask t : all) {
    int k0 = 0, k1 = 0;
    if (t.kind == 2 && chain_batch_of(q, p, t.i, t.c, t.k, &k0, &k1)) {
      if (t.k != k1 - 1) continue;  // absorbed by the batch that ends at column k1 - 1
      t.kind = 6;
      t.k0 = k0;
    }
    out.push_back(t);
  }
  return out;
}
