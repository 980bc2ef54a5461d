"""`imm` — the reference's package name as an alias of `imm_amd` ("same module path/class/ctor"): every
submodule here replaces itself with its imm_amd counterpart in sys.modules, so `from imm.models.imm_model import IMMModel`
(reference scripts/train.py:13) yields the MI355X-backed class and shares its state (e.g. `IMMModel.num_instances`) with
`imm_amd.models.imm_model`.  Nothing is implemented in this package."""
