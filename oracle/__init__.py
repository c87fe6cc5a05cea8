"""CPU oracle for the swarm_loop hot path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(HKUST-Aerial-Robotics/Omni-swarm, ``swarm_loop``) used as the *checker* by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py``.  Nothing in the shipped product path (``omni-swarm_amd/``,
``include/``) may import, link or execute anything from here.

Parity pinning status (see DESIGN.md section "Oracle"):
  * SuperPoint network        -- pinned against the reference's own PyTorch
                                 module (swarm_loop/superpoint.ipynb cell 3)
                                 exec'd in the build container; golden vectors
                                 in tests/golden/ (tools/gen_golden.py).
  * getKeyPoints / NMS2       -- restated from superpoint_tensorrt.cpp:164-310;
                                 literal simulation + characterised form.  The
                                 reference holds no golden vectors: parity
                                 unpinned beyond the literal restatement.
  * computeDescriptors / PCA  -- torch.grid_sample is the reference's own call.
  * faiss IndexFlatIP, cv::BFMatcher, MobileNetVLAD -- third-party, absent from
    /root/reference: PARITY UNPINNED (published algorithm restated).
"""
