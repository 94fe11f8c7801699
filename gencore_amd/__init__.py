"""gencore_amd — MI355X-native consensus-read engine for the Cluster -> Group -> consensus path of OpenGene/gencore.

Only what the path needs lives here: csrc/ (HIP kernels + the C-ABI of include/gencore_amd.h), the ctypes host
mirror (capi, engine), the SoA batch container and the synthetic workload generator.  There is no CPU fallback.
"""
__version__ = "0.1.0"
