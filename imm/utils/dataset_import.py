"""Import-path alias: `imm.utils.dataset_import` IS `imm_amd.utils.dataset_import` (the same module object), so that code written against the reference's
package layout (/root/reference/scripts/train.py:13-19, scripts/test.py, imm/eval/eval_imm.py:14-15) runs unchanged."""
import sys

import imm_amd.utils.dataset_import as _real

sys.modules[__name__] = _real
