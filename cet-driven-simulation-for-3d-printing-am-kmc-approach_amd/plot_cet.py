"""Post-processing of the KMC runs (same CSV contract as the reference ``plot_cet.py``).

Collects ``outputs/impurity_c_*/metrics_*.csv`` (written by kmc_simulation.run_kmc next to
``metrics.csv``) and plots ``AspectRatio``, ``DefectDensity`` and ``EquiaxedFraction`` against
``Step`` (plot_cet.py:26,49,60-62,69-71).  Unlike the reference this is importable without side
effects: call :func:`analyze` or run the file as a script.
"""
import glob
import os


def discover(input_root="outputs"):
    """label -> csv path, one per carbon level (plot_cet.py:25-31)."""
    files = {}
    for path in sorted(glob.glob(os.path.join(input_root, "impurity_c_*", "metrics_*.csv"))):
        level = os.path.basename(os.path.dirname(path)).split("_")[-1]
        files[f"{level}% C"] = path
    return files


def analyze(input_root="outputs", outdir="analysis_plots"):
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    import pandas as pd

    os.makedirs(outdir, exist_ok=True)
    files = discover(input_root)
    print("Found files:", files)
    data = {label: pd.read_csv(path) for label, path in files.items()}
    if not data:
        print("no metrics_*.csv found; run main.py / kmc_simulation.run_kmc first")
        return {}
    for metric, ylabel, fname, logy in (("AspectRatio", "Aspect Ratio", "aspect_ratio_vs_step.png", False),
                                        ("DefectDensity", "Defect Density [a.u.]", "defect_density_vs_step.png", True),
                                        ("EquiaxedFraction", "Equiaxed Fraction", "equiaxed_fraction_vs_step.png", False)):
        plt.figure(figsize=(7, 5))
        for label, df in data.items():
            plt.plot(df["Step"], df[metric], label=label)
        plt.xlabel("Step")
        plt.ylabel(ylabel)
        if logy and all((df[metric] > 0).any() for df in data.values()):
            plt.yscale("log")
        plt.legend()
        plt.grid(True, alpha=0.3)
        plt.tight_layout()
        plt.savefig(os.path.join(outdir, fname))
        plt.close()
    final = pd.DataFrame({label: {"Final AspectRatio": df["AspectRatio"].iloc[-1],
                                  "Final DefectDensity": df["DefectDensity"].iloc[-1],
                                  "Final EquiaxedFraction": df["EquiaxedFraction"].iloc[-1]}
                          for label, df in data.items()}).T
    for metric, ylabel, fname in (("Final AspectRatio", "Aspect Ratio", "bar_aspect_ratio.png"),
                                  ("Final DefectDensity", "Defect Density [a.u.]", "bar_defect_density.png"),
                                  ("Final EquiaxedFraction", "Equiaxed Fraction", "bar_eqfrac.png")):
        plt.figure(figsize=(6, 4))
        final[metric].plot(kind="bar")
        plt.ylabel(ylabel)
        plt.xlabel("Carbon Impurity Level")
        plt.xticks(rotation=0)
        plt.tight_layout()
        plt.savefig(os.path.join(outdir, fname))
        plt.close()
    print(f"Analysis complete. Plots saved in {outdir}/")
    return final


if __name__ == "__main__":
    analyze()
