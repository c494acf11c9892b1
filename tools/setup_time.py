#!/usr/bin/env python3
"""One-off costs of a Stage-5 handle on config D (not part of any bench step): synthetic
generation, calico_rp_add_observations (host copy + checks), the first lm_begin (FR / part /
piece structure on the host + upload), for observations that arrive FR by FR and shuffled.
  python tools/setup_time.py"""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from calico_amd import capi, synth
cfg = synth.preset("D")
t=time.time(); obs = synth.observations(cfg); print("generate %.2f s" % (time.time()-t), obs[0].size)
P = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
P.set_intrinsics(synth.intrinsics(cfg)); P.set_points(synth.points(cfg))
t=time.time(); P.add_observations(*obs); print("add_observations %.2f s" % (time.time()-t))
x0 = synth.initial_guess(cfg)
t=time.time(); P.lm_begin(x0, capi.default_options(max_num_iterations=3)); print("lm_begin (structure + upload) %.2f s" % (time.time()-t))
t=time.time(); P.lm_step(); P.synchronize(); print("first step %.3f s" % (time.time()-t))
t=time.time(); P.lm_step(); P.synchronize(); print("second step %.3f s" % (time.time()-t))
P.lm_end()
# shuffled input order
rng=np.random.default_rng(0); perm=rng.permutation(obs[0].size)
Q = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
Q.set_intrinsics(synth.intrinsics(cfg)); Q.set_points(synth.points(cfg))
sh=[a[perm] for a in obs]
t=time.time(); Q.add_observations(*sh); print("add_observations (shuffled) %.2f s" % (time.time()-t))
t=time.time(); Q.lm_begin(x0, capi.default_options(max_num_iterations=3)); print("lm_begin shuffled (structure + upload) %.2f s" % (time.time()-t))
# two batches, as the reference adds them (calico.cpp:718-729): the second call appends
R = capi.Problem(cfg.cn, cfg.pn, cfg.tn, cfg.p_star_var, cfg.t_star_var, device=0)
R.set_intrinsics(synth.intrinsics(cfg)); R.set_points(synth.points(cfg))
half = obs[0].size // 2
t=time.time(); R.add_observations(*[a[:half] for a in obs]); R.api.num_frs(R.h); R.synchronize(); print("first half: add + structure %.2f s" % (time.time()-t))
t=time.time(); R.add_observations(*[a[half:] for a in obs]); R.api.num_frs(R.h); R.synchronize(); print("second half: add + structure %.2f s" % (time.time()-t))
