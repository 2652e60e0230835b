"""Drop-in for /root/reference/models/hovernet/opt.py: `get_config(nr_type, mode)` (:23-142) -> hover_net_amd.train.get_config
(same phase_list / run_engine shape for the data path; the logging / visualisation callbacks of the reference's config are
host glue and are wired by hover_net_amd.run_engine instead)."""
from hover_net_amd.train import get_config  # noqa: F401
