"""Drop-in for /root/reference/models/hovernet/net_desc.py: `create_model(mode, **kwargs)` (:149-152) and `HoVerNet`
(:14-145) -> hover_net_amd.net_desc (same constructor, attributes, state_dict key set, forward contract)."""
from hover_net_amd.net_desc import HoVerNet, create_model  # noqa: F401
