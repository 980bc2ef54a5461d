"""Import-path alias: `imm.utils.tps_sampler` IS `imm_amd.data.tps` (the same module object), so that code written against the reference's
package layout (/root/reference/scripts/train.py:13-19, scripts/test.py, imm/eval/eval_imm.py:14-15) runs unchanged."""
import sys

import imm_amd.data.tps as _real

sys.modules[__name__] = _real
