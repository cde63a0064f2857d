"""CPU oracle for the SPEC inference hot path  --  TEST INFRASTRUCTURE ONLY.

This package is a from-scratch CPU restatement (PyTorch-CPU fp32 for the trunk / heads,
NumPy float64 variants for error budgeting) of the per-image forward that the reference
composes in ``spec/models/hmr.py:82-122`` and ``camcalib/model.py:72-81``.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the *checker*.  The product path (``spec_amd``) never imports it and
fails loudly when the HIP library is missing.

PARITY STATUS
-------------
* Pinned to the reference's in-tree source: the *composition* (kwarg routing, K[2,2]=0, the
  vfov formula, dict keys, list order of the CamCalib logits, the soft-argmax decode ranges,
  the 49-entry joint map).  ``tests/golden/make_fixtures.py`` imports the reference's own
  ``spec/models/hmr.py``, ``camcalib/model.py``, ``camcalib/cam_utils.py``,
  ``spec/utils/cam_params.py`` and ``spec/constants.py`` over a name shim
  (``oracle/refshim.py``) and the committed ``tests/golden/*.npz`` hold its outputs.
* **Parity unpinned** for the leaf arithmetic that lives in un-vendored dependencies absent
  from ``/root/reference`` and not installable here: ``pare`` (unpinned git HEAD,
  requirements.txt:28: ``resnet50``, ``HMRHead``, ``SMPLCamHead``, ``SMPLHead``, ``SMPL``,
  ``rot6d_to_rotmat``, ``batch_euler2matrix``, ``convert_pare_to_full_img_cam``,
  ``perspective_projection``, ``softargmax1d``) and ``smplx==0.1.28`` (requirements.txt:7:
  ``lbs``, ``batch_rigid_transform``, ``VertexJointSelector``).  Their published algorithms
  are restated here; each function cites the reference call site that constrains it.  The
  reference holds no tests or golden vectors, so closed-form
  known-answer tests in ``tests/test_oracle_kat.py`` stand in for them.
"""
