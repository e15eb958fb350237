"""Flat-name shim for the reference's prot_util module: the rigid-move pieces that do not need BioPython
(move_prot, move_prots, ProtProjection; reference prot_util.py:61-117).  Put diffusion-extensions_amd/compat AND
diffusion-extensions_amd on PYTHONPATH.  pdb_2_rigid_gas / ProtDataset (PDB parsing) are not provided."""
from so3x.se3 import move_prot, move_prots, ProtProjection, ProtData, AffineT  # noqa: F401
from so3x.models import RES_COUNT  # noqa: F401

__all__ = ["move_prot", "move_prots", "ProtProjection", "ProtData", "AffineT", "RES_COUNT"]
