"""icp_amd — MI355X-native ICP / Go-ICP correspondence hot path behind the reference's class API.

  icp_amd.api.Context      Python view of the C ABI (include/icpx.h)
  icp_amd.host             Python handles on the C++ host classes (KDTree, LDT, SingleThreadIcp, GoIcp)
"""
from .lib import IcpxError  # noqa: F401
