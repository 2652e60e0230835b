"""Drop-in for /root/reference/models/hovernet/run_desc.py: `train_step` (:12-109), `valid_step` (:113-167),
`infer_step` (:171-197), `proc_valid_step_output` (:262-333, scalar half) -> hover_net_amd.run_desc.
`viz_step_output` (:201-258, matplotlib / cv2 rendering of a training batch) is host-only visualisation and is not
rebuilt: a config that names it keeps the reference's own function."""
from hover_net_amd.run_desc import infer_step, proc_valid_step_output, train_step, valid_step  # noqa: F401
