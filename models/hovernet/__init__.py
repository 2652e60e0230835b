"""`models.hovernet.*` -- the dotted names the reference's pipeline imports, bound to hover_net_amd (see models/__init__.py)."""
