"""xmap -- drop-in for the hot path of LPD-EPFL-ML/X-MAP on AMD MI355X.

Same module layout as the reference (code/xmap/__init__.py:2-3): the pipeline functions live in
xmap.utils.assist, the tool classes in xmap.core.<module>; the three hot-path pipelines run on the GPU
through libxmap_hip.so (xmap.engine).
"""
__version__ = "0.1.0"
