"""Drop-in for /root/reference/models/hovernet/targets.py: `gen_targets(ann, crop_shape, **kwargs)` (:100-116) and
`gen_instance_hv_map` (:17-96, folded into the same kernels) -> hover_net_amd.targets (hvn_gen_targets on the GPU).  `prep_sample` (:120-145, colour-mapped
visualisation of a sample) is host-only visualisation and is not rebuilt."""
from hover_net_amd.targets import gen_targets, gen_targets_device  # noqa: F401
