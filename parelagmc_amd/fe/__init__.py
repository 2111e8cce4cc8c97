"""Setup-side finite-element builders (meshes, RT0/P0 operators, level hierarchies).

Host-only numpy/scipy code that produces the CSR operators the C-ABI consumes; it stands in
for the MFEM/ParELAG setup calls of the reference drivers and is outside every timed region."""
from .mesh import (Mesh, box_mesh, build_faces, kuhn_cube_tet, mesh_from_json, read_mfem_mesh,  # noqa: F401
                   refine_uniform)
from .problems import (DarcyLevel, DarcyProblem, Hierarchy, SamplerLevel, SamplerProblem,  # noqa: F401
                       build_darcy_problem, build_hierarchy, build_sampler_problem,
                       l2_projection_ops, matern_coefficient)
from .rt0 import build_spaces, mass_contributions, mass_matrix, prolongation_p0  # noqa: F401
from .transfer import box_intersection_gt, l2_projection_hierarchy  # noqa: F401
