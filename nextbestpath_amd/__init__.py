"""nextbestpath_amd -- MI355X (gfx950) native hot path of NextBestPath.

Host side is Python (as the reference is); compute is hand-written HIP behind the C ABI in
``include/nbp_hip.h`` (``libnbp_hip.so``, loaded by ``_lib``).  Sub-packages mirror the
reference's layout for the hot path only: ``networks`` (NBP), ``utility`` (map accumulation,
planner glue), ``simulator`` (camera / depth / raster), ``testers`` (rollout).
"""
__version__ = "0.1.0"
