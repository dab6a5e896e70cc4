"""Tree-search configuration.

Mirrors alphatriangle/config/mcts_config.py:11-77 (same field names, defaults and
``validate_assignment``).  ``to_search_params`` replaces ``to_trimcts_config``: it
produces the parameter block shared by the host search and the HIP tree kernels.
The extra fields (``fpu_parent``, ``virtual_loss``, ``tree_capacity``) cover
behaviour the external search library kept internal.
"""

from __future__ import annotations

from pydantic import BaseModel, ConfigDict, Field

DEFAULT_MAX_SIMULATIONS = 64
DEFAULT_MAX_DEPTH = 8
DEFAULT_CPUCT = 1.5
DEFAULT_MCTS_BATCH_SIZE = 32


class AlphaTriangleMCTSConfig(BaseModel):
    max_simulations: int = Field(default=DEFAULT_MAX_SIMULATIONS, gt=0, description="Simulations per move.")
    max_depth: int = Field(default=DEFAULT_MAX_DEPTH, gt=0, le=64, description="Selection depth cap.")
    cpuct: float = Field(default=DEFAULT_CPUCT, gt=0, description="PUCT exploration constant.")
    dirichlet_alpha: float = Field(default=0.3, ge=0, description="Dirichlet concentration of the root noise.")
    dirichlet_epsilon: float = Field(default=0.25, ge=0, le=1.0, description="Weight of the root noise.")
    discount: float = Field(default=1.0, ge=0.0, le=1.0, description="Discount applied while backing up.")
    mcts_batch_size: int = Field(
        default=DEFAULT_MCTS_BATCH_SIZE, gt=0, description="Leaves gathered per network call (host search)."
    )
    fpu_parent: bool = Field(default=True, description="Unvisited edges take the parent's mean value (else 0).")
    virtual_loss: float = Field(default=-1.0, description="Value charged to in-flight edges when batching leaves.")
    tree_capacity: int | None = Field(default=None, gt=1, description="Nodes per tree; default 2*sims+2.")

    model_config = ConfigDict(validate_assignment=True)

    def capacity(self) -> int:
        return int(self.tree_capacity or (2 * self.max_simulations + 2))

    def to_search_params(self, leaf_batch: int | None = None, module=None):
        """Parameter block for the native searches (host C++ and HIP).  ``module`` names the host build that will
        consume it (``native.host_of(geometry)``: the wide build for boards above 256 cells); default: the 256-cell one."""
        from .. import native

        return (module or native.host()).SearchParams(
            max_simulations=self.max_simulations,
            max_depth=self.max_depth,
            cpuct=self.cpuct,
            dirichlet_alpha=self.dirichlet_alpha,
            dirichlet_epsilon=self.dirichlet_epsilon,
            discount=self.discount,
            leaf_batch=int(leaf_batch if leaf_batch is not None else self.mcts_batch_size),
            fpu_parent=self.fpu_parent,
            virtual_loss=self.virtual_loss,
        )

    # Name kept so code written against the reference keeps working.
    def to_trimcts_config(self):
        return self.to_search_params()


SearchConfiguration = AlphaTriangleMCTSConfig

AlphaTriangleMCTSConfig.model_rebuild(force=True)
