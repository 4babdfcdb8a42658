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

// tasks of step k (column k is made final, everything right of it receives its update) in closed form:
//   xsolve(k+1), diag(k+1)             if block k+1 is factored by this launch (k + 1 < ce)
//   solve(i, k), i = r0 .. R-1         r0 = k + 2 while xsolve(k+1) owns tile (k+1, k)
//   update(i, c, k), c = k+1 .. nblk-1, i = c .. R-1, without (k+1, k+1, k) = diag(k+1)'s fold; the first of them,
//                                      tile (k+2, k+1), as CHAIN_CRIT_PARTS tasks
struct ChainStep {
  int nd, r0, ns, a, b, crit, nu;
};
CHAIN_HD inline ChainStep chain_step(int k, int R, int nblk, int ce) {
  ChainStep s;
  s.nd = k + 1 < ce ? 1 : 0;
  s.r0 = s.nd ? k + 2 : k + 1;
  s.ns = R - s.r0 > 0 ? R - s.r0 : 0;
  s.a = k + 1;
  s.b = nblk - 1;
  s.crit = (s.a <= s.b && R - (s.a + 1) >= 1) ? CHAIN_CRIT_PARTS - 1 : 0;
  s.nu = (s.a <= s.b ? (s.b - s.a + 1) * R - (s.a + s.b) * (s.b - s.a + 1) / 2 - 1 : 0) + s.crit;
  return s;
}

// tickets of a launch over block columns [cb, ce) of a panel with R row tiles and nblk block columns.  A continuation
// launch (cb > 0) starts with diag(cb): tile (cb, cb-1) is final since the launch before; a panel's very first block
// (cb == 0) is factored in front of the launch.
CHAIN_HD inline int64_t chain_task_count(int R, int nblk, int cb, int ce) {
  int64_t tasks = cb > 0 ? 1 : 0;
  for (int k = cb; k < ce; ++k) {
    const ChainStep s = chain_step(k, R, nblk, ce);
    tasks += 2 * s.nd + s.ns + s.nu;
  }
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
    const ChainStep s = chain_step(k, R, nblk, ce);
    if (t >= 2 * s.nd + s.ns + s.nu) {
      t -= 2 * s.nd + s.ns + s.nu;
      continue;
    }
    if (s.nd) {  // xsolve(k+1) in front of diag(k+1): the diagonal task follows the solve of its tile
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
    for (int c = s.a; c <= s.b; ++c) {
      const int i0 = c == s.a ? c + 1 : c;
      const int cnt = R - i0;
      if (c == s.a && s.crit) {
        if (t < CHAIN_CRIT_PARTS) {
          task.kind = 4;
          task.i = i0;
          task.c = c;
          task.k = k;
          task.part = t;
          return task;
        }
        t -= CHAIN_CRIT_PARTS - 1;
      }
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
