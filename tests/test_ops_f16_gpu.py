"""Every kernel test of tests/test_ops_gpu.py again with fp16 ELEMENTS, i.e. against libctrlv_hip_f16.so (the same
sources built with -DCTRLV_ELEM_F16: csrc/common.h): same cases, same fp32 PyTorch references on the same (fp16-rounded)
inputs, element-output bounds scaled to fp16's rounding floor (`tol()` in that file: a sixth of the bf16 bound).

The source of test_ops_gpu.py is executed in THIS module's namespace with EL = torch.float16, so that every test function
defined there exists here bound to the fp16 element type -- one set of cases to maintain.
"""
import os

import torch

_SRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "test_ops_gpu.py")
exec(compile(open(_SRC).read(), _SRC, "exec"), globals())      # defines pytestmark (gpu), the fixtures and the tests
EL = torch.float16                                               # noqa: F811  (read by the tests at call time)
