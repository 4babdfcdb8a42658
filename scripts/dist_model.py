"""Critical-path model of the block-column driver at G GPUs from a WORLD-SIZE-1 kernel timeline.

usage: dist_model.py <timeline.csv from scripts/timeline.py> <n> <nb> [link_GBps=120] [update_TFLOPs=62] [chain_TFLOPs=45]

Round 6: with the persistent chain (one launch per panel chunk, tile tasks that wait inside the launch) the SPAN of a
chain at world size 1 is no longer its cost -- it runs beside the whole trailing update of the step before and takes as
long as that update does (226 ms at the top of config 4).  The model therefore prices the chain of panel k from its WORK:
  chain_k = max(blocks x 36 us (the diagonal chain's period, profiles/r04_b), rows_k nb^2 flops / chain_TFLOPs) + packs
(chain_TFLOPs = 45: what the chain's tile tasks sustain on an otherwise idle chip, profiles/r06_a), and prints the measured
span beside it.  The broadcast of panel k is priced at bytes_k / link + 50 us: every receiver takes the panel over ONE xGMI
link from the root (direct sends use one link per peer; a ring is bound by one link as well); 120 GB/s = 0.78 of the 153
GB/s a link peaks at is RCCL's usual large-message efficiency, 60 and 40 GB/s are the pessimistic rows.

Per panel k the timeline gives, on the one GPU: gate_k (panel k-1 applied to block column k, priority stream),
chain_k (first potf2 of the panel .. end of its last pack kernel: potf2 / trsm / in-panel updates / packs) and the
packs alone.  The update of step k costs flops_k = nb * rows_{k+1}^2 at `update_TFLOPs` (measured rate of the trailing
update at that size) and is divided by G; the broadcast of panel k moves bytes_k = 8 rows_k nb over one xGMI link per
receiver (direct sends from the root use one link per peer; a ring is bound by one link as well).

Two streams carry the factorisation (tinygp_amd/distributed.py):
  pipeline (priority stream + RCCL):  P_k = gate_k + chunked(chain_k, bcast_k)   one after the other, panel by panel;
      chunked(c, b) = max(c, b) + min(c, b) / nch: the column chunks of a panel are sent while the chain factors the
      next ones (nch = 4 above 128 MB per panel, 2 above 64 MB, else 1)
  updates (main stream):               U_k = flops_k / (G * rate) (+ the forward step, hidden on the update stream)
With the chain pipeline two panels ahead of the updates (three ring slots) the evaluation takes between
  max(sum P_k, sum U_k) + P_0   (perfect overlap)   and   sum_k max(P_k, U_k) + P_0   (depth-1 behaviour)."""
import sys

path, n, nb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
link = float(sys.argv[4]) if len(sys.argv) > 4 else 120.0
rate = float(sys.argv[5]) if len(sys.argv) > 5 else 62.0
chain_rate = float(sys.argv[6]) if len(sys.argv) > 6 else 45.0
rows = []
with open(path) as f:
    f.readline()
    for line in f:
        a = line.rstrip("\n").split(",", 2)
        b = a[2].rsplit(",", 7)
        rows.append((float(a[0]), float(a[1]), b[0], b[1]))
# the LAST evaluation in the trace: from the first assembly launch behind the previous evaluation's last reduction
# (sum_squares closes an evaluation) -- or from the very first assembly launch
asm = [i for i, r in enumerate(rows) if r[2].startswith(("kmat_fast", "kmat_kernel"))]
ends = [i for i, r in enumerate(rows) if r[2].startswith("sum_squares")]
prev_end = ends[-2] if len(ends) >= 2 else -1
start = next(i for i in asm if i > prev_end)
ev = rows[start:]
t0 = ev[0][0]
potf2 = [i for i, r in enumerate(ev) if r[2].startswith("potf2")]
nblk, per = n // nb, nb // 128
if len(potf2) == nblk:  # the persistent chain (round 4 on): ONE potf2 launch per panel, the other blocks are chain tasks
    per = 1
assert len(potf2) == nblk * per, (len(potf2), nblk * per)
pq = ev[potf2[1]][3]  # the priority stream's queue id (second potf2 of panel 0 runs there)
out = []
for k in range(nblk):
    lo = ev[potf2[k * per]][0]
    hi = ev[potf2[(k + 1) * per]][0] if k + 1 < nblk else ev[-1][1]
    packs = [r for r in ev if ("pack_panel" in r[2] or r[2].strip() == "void") and lo <= r[0] < hi]  # ("void": timeline.py cut the name of a kernel in a global anonymous namespace)
    chain_end = max(r[1] for r in packs)
    gate = [r for r in ev if r[2].startswith("gemm_nt") and r[3] == pq and r[1] <= lo + 1 and r[0] >= (out[-1]["chain_end"] if out else 0)]
    gate = [g for g in gate if g[1] - g[0] > 0]
    g_us = (gate[-1][1] - gate[-1][0]) if (gate and k > 0) else 0.0
    out.append({"k": k, "rows": n - k * nb, "chain": chain_end - lo, "pack": sum(r[1] - r[0] for r in packs), "gate": g_us,
                "chain_end": chain_end})
total = ev[-1][1] - t0
print(f"# world size 1, N = {n}, nb = {nb}: one evaluation {total / 1e3:.1f} ms, {len(ev)} kernels")
print("#   k    rows   gate_us  chain_us (span at world size 1)  chain_us (model)  (pack_us)   bytes_MB   bcast_us@%g GB/s   update_ms@1GPU" % link)
for o in out:
    o["bytes"] = 8.0 * o["rows"] * nb
    o["bcast"] = o["bytes"] / (link * 1e3) + 50.0  # us
    o["span"] = o["chain"]
    o["chain"] = max((nb // 128) * 36.0, o["rows"] * float(nb) * nb / (chain_rate * 1e12) * 1e6) + o["pack"]
    m = o["rows"] - nb
    o["upd"] = nb * float(m) * m / (rate * 1e12) * 1e6  # us at 1 GPU
    if o["k"] % max(1, nblk // 16) == 0 or o["k"] == nblk - 1:
        print(f"  {o['k']:4d} {o['rows']:7d} {o['gate']:9.0f} {o['span']:14.0f} {o['chain']:26.0f}  ({o['pack']:7.0f}) {o['bytes'] / 1e6:10.1f} {o['bcast']:12.0f} {o['upd'] / 1e3:14.2f}")


def nch(b):
    return 4 if b >= 128e6 else 2 if b >= 64e6 else 1


print("# G   sum P (pipeline) ms   sum U (updates) ms   T_lower ms   T_upper ms   speed-up vs the model's own G = 1 (lower .. upper bound of T)")
t_one = None
for G in (1, 2, 4, 8):
    P, U, up = [], [], 0.0
    for o in out:
        b = o["bcast"] if G > 1 else 0.0
        c = o["chain"]
        P.append(o["gate"] + max(c, b) + min(c, b) / nch(o["bytes"]))
        U.append(o["upd"] / G)
    lower = max(sum(P), sum(U)) + P[0]
    upper = sum(max(p, u) for p, u in zip(P[1:] + [0.0], U)) + P[0]
    t_one = lower if t_one is None else t_one
    print(f"  {G}   {sum(P) / 1e3:12.1f} {sum(U) / 1e3:20.1f} {lower / 1e3:14.1f} {upper / 1e3:12.1f}      {t_one / lower:.2f} .. {t_one / upper:.2f}")
    if G == 8:
        # where the schedule stops being update-bound: the first panel whose pipeline step outlasts its (shared) update
        first = next((o["k"] for o, pk, uk in zip(out, P, U) if pk > uk), None)
        bb = next((o["k"] for o in out if o["bcast"] > o["chain"]), None)
        tail = sum(max(pk - uk, 0.0) for pk, uk in zip(P, U)) / 1e3
        print(f"# G = 8: the pipeline step outlasts the update from panel {first} on ({n - (first or 0) * nb} rows left); exposed "
              f"pipeline time summed {tail:.1f} ms; the broadcast outlasts the chain (broadcast-bound) from panel {bb} on"
              if first is not None else "# G = 8: update-bound throughout")
