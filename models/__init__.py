"""Import-by-name drop-in package: the reference resolves its model plugins with
`import_module("models.hovernet.net_desc")` etc. (/root/reference/infer/base.py:61-77,
/root/reference/run_train.py:158-190 via models/hovernet/opt.py).  With this repository's root ahead of the
reference's on `sys.path`, those imports land on the MI355X-native path in `hover_net_amd/`."""
