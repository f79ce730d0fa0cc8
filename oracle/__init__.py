"""CPU oracle for the DPoser diffusion hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dposer_amd/`` imports this package; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may use it,
and there only as the checker / the timed CPU baseline -- never as the thing shipped.

Parity status
-------------
* ``oracle.score_ref``  -- restatement of the reference's score network, sub-VP/VP/VE SDE scalars,
  DSM loss, Adam/clip/EMA train step, Euler-Maruyama / Langevin sampler steps and the DPoser prior
  loss.  PINNED: checked against golden vectors captured from the imported reference
  (``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``, ``tests/test_oracle_golden.py``).
* ``oracle.task_loops`` -- restatement of the task loops ``DPoserComp.optimize`` (run/completion.py:167-207) and
  ``MotionDenoise.optimize`` (run/motion_denoising.py:199-300).  PINNED to outputs of the reference's OWN loops (goldens
  ``g14``, ``g15``; the motion-denoising loop drove ``oracle.fk_torch`` as its body model, so the loop is pinned, not the LBS).
* ``oracle.fk_ref``     -- restatement of ``smplx==0.1.28`` ``lbs.py`` (un-vendored third-party
  dependency, absent from /root/reference, not installable here) and of the reference's own
  ``rot6d_to_mat3x3``.  rot6d is pinned by a golden vector; the smplx LBS half is
  **parity unpinned** (no smplx wheel, no SMPL-X asset, the reference holds no test for it).
* ``oracle.philox``     -- numpy Philox4x32-10 + the bit->float maps the HIP kernels use, so tests
  can inject the *same* random numbers into the oracle that the kernels draw on the GPU.
"""
