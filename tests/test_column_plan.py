"""csrc/sri_column_plan.h on the CPU: which tile geometry takes which columns of a panel.

The planner is pure C++ (a cost model over the tile counts of each candidate launch, no device call); the library's run()
feeds it the handle's geometries and follows its choice.  tests/cpp/column_plan_check.cpp checks the choices at the panel
sizes that were measured on the MI355X (profiles/r04_geometry_by_columns.log), the model against those measurements, and —
over 200 000 random shapes — the invariants run() relies on: kept columns are whole tiles of every primary geometry, an
alternate is named exactly when columns are left over, and the plan never deviates from the operator's own geometry for
less than 5 % of modelled gain.  The same header holds the ROW side of the launch plan — the operator's own geometry by
rows and the parts it is cut into (create_from_matrix follows it): checked at the sizes DESIGN.md names and, for every row
count up to 4 200, that the parts tile the rows without gap or overlap, start on row blocks, never pad more than the whole
operator would and never more than 63 rows.  What the planned launches compute is tests/test_apply_gpu.py's subject
(test_column_plan_gives_the_bits_of_the_own_geometry, test_parts_give_the_bits_of_the_whole_operator)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "experimental_gpu_programming_for_a_spectral_numerical_integration_amd", "csrc")
SRC = os.path.join(ROOT, "tests", "cpp", "column_plan_check.cpp")


def test_column_plan_choices_and_invariants(tmp_path):
    exe = os.path.join(str(tmp_path), "column_plan_check")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-O2", "-fsanitize=undefined", "-fno-sanitize-recover=all",
                           "-I", CSRC, SRC, "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "column_plan_check: 0 failures" in out.stdout
