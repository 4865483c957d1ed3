"""CPU oracle for the TG-Pose point-cloud forward hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, on the CPU, the algorithm of the reference path so that the HIP
implementation can be checked against it:

    oracle/csrc/tgp_oracle.c   exact-arithmetic kNN / 1-NN / Chamfer (C, gcc)
    oracle/gcn_ref.py          graph ops of network/fs_net_repo/gcn3d.py (torch CPU ops)
    oracle/posenet_ref.py      Face_Enc / PH_Predictor / Face_Dec / heads / PoseNet9D.forward
    oracle/loss_ref.py         chamfer_3DDist forward/backward and calc_dcd
    oracle/eval_ref.py         pairwise 3D IoU / pose error of the NOCS mAP
    oracle/tda_loss_ref.py     the TDA loss bundle
    oracle/input_ref.py        depth frame + mask + box -> cloud (evaluation loader input side; the two OpenCV calls it
                               restates are the one piece whose parity is unpinned: cv2 is not installable here)

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it -- as the checker, never as the thing measured or shipped.  The product package
(``tg-pose_amd/``) never imports from here and fails loudly when its HIP library is missing.

Parity pinning: the restatement is checked against outputs of the reference itself, imported
unmodified from /root/reference in the build container by ``tests/golden/make_golden.py``; the
resulting vectors are committed under ``tests/golden/`` (the reference ships no golden vectors
of its own: SURVEY.md section 4).
"""
