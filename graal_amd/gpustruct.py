"""``GPUStruct``-compatible view of the fragment layout held by the engine.

The reference's ``gpustruct.GPUStruct`` (``gpustruct.py:9-211``) packs 14 PyCUDA allocations into a device struct and
mirrors them as numpy attributes after ``copy_from_gpu()``; callers (``simulation_loader.py:109-115,782``,
``pyramid_sparse.py:1432-1457``) only ever use ``copy_from_gpu()`` / ``copy_to_gpu()`` and the attributes.  Here the
device side lives behind the C-ABI handle; this class keeps the same method and attribute names.
"""
import numpy as np

from .lib import FIELDS


class GPUStruct(object):
    def __init__(self, engine, soa):
        self._engine = engine
        for k in FIELDS:
            setattr(self, k, np.array(soa[k], dtype=np.int32, copy=True))

    def as_dict(self):
        return {k: getattr(self, k) for k in FIELDS}

    def copy_to_gpu(self, skip=None):
        self._engine.upload_frags(self.as_dict())

    def copy_from_gpu(self, skip=None):
        self._engine.download_frags(self.as_dict())

    def get_ptr(self):
        raise RuntimeError("the device struct is owned by libgraal_hip.so; use the sampler / Engine methods")

    def __str__(self):
        return "".join("%s: %s\n" % (k, str(getattr(self, k))) for k in FIELDS)
