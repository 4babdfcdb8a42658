// chain_tasks.h -- the persistent panel chain's ticket -> task map, shared by the kernel (chol.hip, chain_kernel), the
// host side that sizes the launch (launch_chain) and the C ABI's test hook (tgp_chain_task: tests/test_chain_tasks.py
// enumerates every ticket of a launch on the CPU and checks that each task exists exactly once and waits for EARLIER
// tickets only -- what makes one-task-per-workgroup deadlock-free without co-residency).  CHAIN_HD is defined by the
// includer (__host__ __device__ in HIP code, empty in a host-only translation unit).
#pragma once
#include <cstdint>

struct ChainTask {
  int kind;  // 0 solve(i, c) | 1 diag(c) | 2 update(i, c, k) | 3 update of the diagonal tile (c, c) from column k |
             // 4 one of CHAIN_CRIT_PARTS parts (`part`) of update(i, c, k) | 5 xsolve(c): the streamed solve of tile (c, c-1) | -1 none
  int i, c, k, part;
};
constexpr int CHAIN_CRIT_PARTS = 8;  // workgroups that share the update of tile (k+2, k+1) from column k

// Tasks of step k (column k is made final, everything right of it receives its update) and their TICKET ORDER.
//   solve(i, k), i = r0 .. R-1            r0 = k + 2 while xsolve(k+1) owns tile (k+1, k)
//   U1: update(i, k+1, k), i = k+2 ..     the column the NEXT step solves; its first tile, (k+2, k+1), as CHAIN_CRIT_PARTS tasks
//   D:  update(k+2, k+2, k)               the diagonal tile the step after next factors
//   A:  xsolve(k+2), diag(k+2)            if block k+2 is factored by this launch (k + 2 < ce) -- THE NEXT STEP'S diagonal tasks
//   U2: update(i, c, k), c = k+2 .. nblk-1, i = c .., without D        the bulk
// (the launch opens with diag(cb) if cb > 0, then xsolve(cb+1), diag(cb+1) if cb + 1 < ce.)
// Round 5: A sits IN FRONT of the bulk of its predecessor step.  Until round 4 a step was [A | solves | all updates], so the
// diagonal tasks of step k+1 drew their tickets behind the (R-k)^2/2 bulk updates of step k -- ~460 tasks of 22 us on 256
// compute units at N = 4 096 -- although they need none of them: the stamped timeline showed xsolve / diag STARTING 15-37 us
// late and the first ~11 blocks of a 32-block launch at 45-65 us per block instead of 36 (profiles/r05_p).  Everything A
// waits for -- the updates of tiles (k+2, k+1) and (k+2, k+2) through column k, the progress of diag(k+1) -- has an earlier
// ticket in this order too (tests/test_chain_tasks.py checks every wait of every task of a launch).
struct ChainStep {
  int nd0;   // the launch's FIRST step only: xsolve(k+1), diag(k+1) in front (0 / 1)
  int r0, ns;
  int a, b;  // updated columns a = k+1 .. b = nblk-1 (a > b: none)
  int crit;  // extra tickets of the split update of tile (k+2, k+1) (CHAIN_CRIT_PARTS - 1, or 0)
  int nu1;   // U1 tickets (crit included)
  int nD;    // D (0 / 1)
  int nA;    // A: 2 or 0
  int nu2;   // U2 tickets
};
CHAIN_HD inline ChainStep chain_step(int k, int R, int nblk, int cb, int ce) {
  ChainStep s;
  const int nd = k + 1 < ce ? 1 : 0;  // block k+1 is factored by this launch (its diagonal tasks precede this step's solves)
  s.nd0 = (k == cb) ? nd : 0;
  s.r0 = nd ? k + 2 : k + 1;
  s.ns = R - s.r0 > 0 ? R - s.r0 : 0;
  s.a = k + 1;
  s.b = nblk - 1;
  const bool upd = s.a <= s.b;
  s.crit = (upd && R - (s.a + 1) >= 1) ? CHAIN_CRIT_PARTS - 1 : 0;
  s.nu1 = upd ? (R - (s.a + 1) > 0 ? R - (s.a + 1) : 0) + s.crit : 0;
  s.nD = (s.a + 1 <= s.b) ? 1 : 0;  // (nblk <= R: row k+2 exists whenever column k+2 does)
  s.nA = (k + 2 < ce) ? 2 : 0;
  int u2 = 0;
  for (int c = s.a + 1; c <= s.b; ++c) u2 += R - c;
  s.nu2 = u2 - s.nD;
  return s;
}
CHAIN_HD inline int chain_step_tickets(const ChainStep& s) { return 2 * s.nd0 + s.ns + s.nu1 + s.nD + s.nA + s.nu2; }

// tickets of a launch over block columns [cb, ce) of a panel with R row tiles and nblk block columns.  A continuation
// launch (cb > 0) starts with diag(cb): tile (cb, cb-1) is final since the launch before; a panel's very first block
// (cb == 0) is factored in front of the launch.
CHAIN_HD inline int64_t chain_task_count(int R, int nblk, int cb, int ce) {
  int64_t tasks = cb > 0 ? 1 : 0;
  for (int k = cb; k < ce; ++k) tasks += chain_step_tickets(chain_step(k, R, nblk, cb, ce));
  return tasks;
}

CHAIN_HD inline ChainTask chain_decode_ticket(int t, int R, int nblk, int cb, int ce) {
  ChainTask task = {-1, 0, 0, 0, 0};
  if (cb > 0) {
    if (t == 0) {
      task.kind = 1;
      task.c = cb;
      return task;
    }
    --t;
  }
  for (int k = cb; k < ce; ++k) {
    const ChainStep s = chain_step(k, R, nblk, cb, ce);
    const int all = chain_step_tickets(s);
    if (t >= all) {
      t -= all;
      continue;
    }
    if (s.nd0) {  // the launch's first step: xsolve(k+1) in front of diag(k+1)
      if (t < 2) {
        task.kind = t == 0 ? 5 : 1;
        task.c = k + 1;
        return task;
      }
      t -= 2;
    }
    if (t < s.ns) {
      task.kind = 0;
      task.i = s.r0 + t;
      task.c = k;
      return task;
    }
    t -= s.ns;
    if (t < s.nu1) {  // U1: column a from column k; its first tile (a+1, a) in CHAIN_CRIT_PARTS parts
      const int i0 = s.a + 1;
      task.c = s.a;
      task.k = k;
      if (s.crit) {
        if (t < CHAIN_CRIT_PARTS) {
          task.kind = 4;
          task.i = i0;
          task.part = t;
          return task;
        }
        t -= CHAIN_CRIT_PARTS - 1;
      }
      task.kind = 2;
      task.i = i0 + t;
      return task;
    }
    t -= s.nu1;
    if (t < s.nD) {  // D: the diagonal tile (k+2, k+2) from column k
      task.kind = 3;
      task.i = task.c = s.a + 1;
      task.k = k;
      return task;
    }
    t -= s.nD;
    if (t < s.nA) {  // A: the NEXT step's diagonal tasks
      task.kind = t == 0 ? 5 : 1;
      task.c = k + 2;
      return task;
    }
    t -= s.nA;
    for (int c = s.a + 1; c <= s.b; ++c) {  // U2: the bulk
      const int i0 = (c == s.a + 1) ? c + 1 : c;  // (tile (a+1, a+1) was D)
      const int cnt = R - i0;
      if (t < cnt) {
        task.i = i0 + t;
        task.c = c;
        task.k = k;
        task.kind = task.i == c ? 3 : 2;
        return task;
      }
      t -= cnt;
    }
    return task;  // (unreachable for t < chain_task_count)
  }
  return task;
}
