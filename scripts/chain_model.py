"""Critical-path model of the persistent panel chain's diagonal (chain-bound regime: few rows left, an idle chip), from the
phase durations the device stamps give (profiles/r04_b).  Events per 128-column block c, times in us:

  T[c]      L_cc final (potf2(c) done and published)
  X[c]      X_{c,c-1} published (the solve of tile (c, c-1) done)
  tile[c]   tile (c, c-1) carries its last update, from column c-2:  X[c-1] + t_update
  the solve of tile (c, c-1) needs the tile and the column blocks of L_{c-1,c-1}; with progress flags block j of 8 is
  in memory at T[c-1] - (7 - j) * t_step, without them everything at T[c-1]
  potf2(c) starts when the fold of X_{c,c-1} is complete; T[c] = start + t_potf2 + t_pub

It reproduces the measured periods of the versions of round 4 and prices the next steps.
  python scripts/chain_model.py"""


def period(*, t_potf2, t_update, t_solve, t_fold, streamed, two_units, t_step=3.2, t_pub=0.6, t_hand=1.5, blocks=40):
    """steady-state period; t_solve / t_fold = the MFMA-bound durations of the whole solve / fold on one compute unit"""
    T = [0.0, 0.0]
    X = [0.0, 0.0]
    for c in range(2, blocks):
        tile = X[c - 1] + t_update
        if not streamed:  # everything behind the FINAL flag
            start = max(tile, T[c - 1] + t_hand)
            x_done = start + t_solve
            fold_done = x_done + t_fold
        else:
            # column block j of the solve can start at max(its data, the previous block done)
            per = (t_solve if two_units else t_solve + t_fold) / 8.0
            t = tile + t_hand
            for j in range(8):
                ready = T[c - 1] - (7 - j) * t_step + (t_hand if j == 7 else 0.5)
                t = max(t, ready) + per
            x_done = t
            fold_done = x_done + (t_fold / 8.0 + 3.0 if two_units else 0.0)  # the fold lags by one block + hand-off
        X.append(x_done + 0.3)
        T.append(fold_done + 0.5 + t_potf2 + t_pub)
    return T[-1] - T[-2]


if __name__ == "__main__":
    base = dict(t_potf2=24.3, t_update=22.5, t_solve=10.8, t_fold=12.4, streamed=False, two_units=False)
    rows = [
        ("v3  everything behind the FINAL flag                         (measured 50.5)", base),
        ("v5  streamed behind progress flags, one workgroup            (measured 49.9)",
         dict(base, t_potf2=27.0, t_solve=13.0, t_fold=13.0, streamed=True)),
        ("v6  + the critical update over 4 workgroups (10 us)          (measured 39.5)",
         dict(base, t_potf2=27.0, t_solve=13.0, t_fold=13.0, streamed=True, t_update=10.0)),
        ("v7  + solve and fold on two compute units                    (measured 35.5)",
         dict(base, t_potf2=27.0, t_solve=13.0, t_fold=13.0, streamed=True, t_update=10.0, two_units=True)),
        ("v8  + the critical update over 8 workgroups (7.5 us)         (measured ~34.5)",
         dict(base, t_potf2=27.0, t_solve=13.0, t_fold=13.0, streamed=True, t_update=7.5, two_units=True)),
        ("next: the stream on the 4x4x4 MFMA form (8 + 8 us)",
         dict(base, t_potf2=27.0, t_solve=8.0, t_fold=8.0, streamed=True, t_update=7.5, two_units=True)),
        ("next: potf2 20 us (its eight elimination steps are the chain now)",
         dict(base, t_potf2=20.0, t_solve=13.0, t_fold=13.0, streamed=True, t_update=7.5, two_units=True, t_step=2.4)),
        ("next: both",
         dict(base, t_potf2=20.0, t_solve=8.0, t_fold=8.0, streamed=True, t_update=7.5, two_units=True, t_step=2.4)),
    ]
    for name, kw in rows:
        print(f"{name:84s} model {period(**kw):5.1f} us per block")
