"""plot_cet CSV contract: the columns it needs exist in the CSV layout run_kmc writes."""
import numpy as np
import pandas as pd

import plot_cet


def test_discover_and_analyze(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    cols = ["Step", "Time", "AspectRatio", "EquiaxedFraction", "NucleationDensity", "DefectDensity", "AvgGrainSize",
            "GrainCount", "W_Count", "Re_Count", "C_Count", "NucleationCount", "G_over_R", "G_phys", "R_phys",
            "G_over_R_phys", "CET_Class", "CET_Detected"]                       # kmc_simulation.py:359-378
    for lvl in (0, 10):
        d = tmp_path / "outputs" / f"impurity_c_{lvl}"
        d.mkdir(parents=True)
        df = pd.DataFrame({c: np.arange(3) + 1.0 for c in cols})
        df.to_csv(d / f"metrics_{lvl}.csv", index=False)
    files = plot_cet.discover("outputs")
    assert list(files) == ["0% C", "10% C"]
    final = plot_cet.analyze("outputs", str(tmp_path / "plots"))
    assert list(final.columns) == ["Final AspectRatio", "Final DefectDensity", "Final EquiaxedFraction"]
    assert (tmp_path / "plots" / "aspect_ratio_vs_step.png").exists()
