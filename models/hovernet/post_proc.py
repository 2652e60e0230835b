"""Drop-in for /root/reference/models/hovernet/post_proc.py: `process(pred_map, nr_types=None, return_centroids=False)`
(:94-186) -> hover_net_amd.post_proc.process (instance separation and the instance table on the GPU, contours on the host).
The function is picklable by reference (module-level), as infer/tile.py:137 needs for its worker pool; a worker process that
calls it must own a GPU context (one process per GPU here, no CPU pool)."""
from hover_net_amd.post_proc import process  # noqa: F401
