"""jlama_amd -- host side of jlama-hip, the MI355X (gfx950) tensor backend for Jlama.

The product is the C-ABI library ``jlama_amd/lib/libjlamahip.so`` (sources: ``jlama_amd/csrc``,
contract: ``include/jlama_hip.h``).  This package mirrors, above that ABI, the reference interfaces of the
hot path so parity tests read like the reference's own:

* :class:`jlama_amd.hip_tensor_operations.HipTensorOperations` -- ``TensorOperations``
  (jlama-core/.../tensor/operations/TensorOperations.java:25-161)
* :class:`jlama_amd.model.HipLlamaModel` -- ``AbstractModel.generate()/forward()/sample()``
  (jlama-core/.../model/AbstractModel.java:253-646) on the device-resident Tier-2 API
* :mod:`jlama_amd.jq4` -- the JQ4 weight format (Q4ByteBufferTensor / Q8ByteBufferTensor layouts)
* :mod:`jlama_amd.kv` -- KvBufferCache page geometry
* :mod:`jlama_amd.distributed` -- DistributedContext layer sharding, one process per GPU

There is no CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from . import _native  # noqa: F401

__all__ = ["_native"]
