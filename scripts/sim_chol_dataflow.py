#!/usr/bin/env python
"""Discrete-event model of a DATAFLOW schedule for the blocked Cholesky of the reduced camera system: one
persistent launch, tasks = blocks (r, c) of L drawn column-major from a ticket counter, left-looking (a task applies the updates of
the columns j < c as their ready flags come up, then factors its stacked panel).  Times in microseconds: t_u = one update (two block
loads + 110 fp64 MFMAs per wave), t_p = stacked panel, t_post = store + release, t_flag = flag latency.  Prints the makespan for
47 block columns (512 keyframes) -- the look-ahead schedule in the library takes 1035 us (profiles/r04_h_chol_step_durations.txt).
Not built: see DESIGN.md."""
import heapq, sys
def sim(nbk=47, W=256, t_u=5.5, t_p=4.5, t_post=1.0, t_flag=1.0, t_load=1.5, t_take=1.0):
    nbrows = nbk + 1
    tasks = [(r, c) for c in range(nbk) for r in range(c + 1, nbrows)]
    done = {}                      # (r,c) -> time L_rc visible
    workers = [0.0] * W
    heapq.heapify(workers)
    # tasks must be processed in ticket order by the earliest-free worker; dependency times are known since deps have earlier tickets
    for (r, c) in tasks:
        t = heapq.heappop(workers) + t_take
        for j in range(c):
            ready = max(done[(r, j)], done[(c, j)]) + t_flag
            t = max(t, ready) + t_u      # (load after flag + mfma; t_u includes the L2 latency unless prefetched)
        t += t_p + t_post
        done[(r, c)] = t
        heapq.heappush(workers, t)
    return max(done.values()), [done[(c + 1, c)] for c in range(0, nbk, 6)]
for t_u in (5.5, 4.0, 3.5):
    for W in (256, 512):
        tot, marks = sim(t_u=t_u, W=W)
        print("t_u %.1f W %d: total %.0f us" % (t_u, W, tot), ["%.0f" % m for m in marks])
