"""IcpUsingPointMatcher chain (the loop-closure matcher, back_end/loop_detector.cc:304) on a 120 k-point pair: the device
chain (2 uploads, device sampling / CalculateNormals / score pass) against the host-side chain it replaced."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import staticmapping_amd as sm
from staticmapping_amd import synth
a, b, T = synth.scan_pair("cfg2", n_points=120000)
guess = synth.make_pose(t=(0.6, 0, 0))
for dev_chain in (False, True):
    m = sm.IcpPointMatcherHip(max_points=131072, prob=0.9, seed=1, device_chain=dev_chain)
    m.set_input_source(b); m.set_input_target(a)
    m.align(guess)
    t = time.time(); reps = 5
    for _ in range(reps): ok, R = m.align(guess)
    dt = (time.time() - t) / reps
    print(f"IcpUsingPointMatcher chain, device_chain={int(dev_chain)}: {dt*1e3:.2f} ms per Align (iterations {m.iterations}, score {m.get_fitness_score():.4f}, ok={ok}) err={sm.se3_error(R, T)}")
    m.close()
