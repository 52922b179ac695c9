"""utils.rrm_utils -- call-surface shim.

In the reference this module is an unreferenced older duplicate of the DenseEnergyLoss code
(utils/rrm_utils.py:352-416 vs utils/seg_helper.py:191-208,864-903).  The north star
asks that the name stays importable, so it re-exports the live implementation.
"""
from .seg_helper import DenseEnergyLoss, DenseEnergyLossFunction  # noqa: F401
