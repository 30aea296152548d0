"""
Feature-family column layouts shared by the HIP kernels (aliby_amd/csrc/feat_*.hip) and the host.

Key names follow what the reference's tables show for cp_measure output
(examples/01_cell_painting_tiff.py:159-162: "0/max/intensity/Intensity_IntegratedIntensity") and
CellProfiler's published measurement names; cp_measure 0.1.17 itself is not in the container, so the
exact spelling of keys beyond that pattern is "parity unpinned" (SURVEY.md §8c).  The column-count
identity 4 + 6*S + 5*I + 10*P = 632 quoted at examples/01:156-158 is reproduced with S=78, I=16, P=8.
"""

from __future__ import annotations

INTENSITY_CORE = [
    "Intensity_IntegratedIntensity",
    "Intensity_MeanIntensity",
    "Intensity_StdIntensity",
    "Intensity_MinIntensity",
    "Intensity_MaxIntensity",
]
INTENSITY_EDGE = [
    "Intensity_IntegratedIntensityEdge",
    "Intensity_MeanIntensityEdge",
    "Intensity_StdIntensityEdge",
    "Intensity_MinIntensityEdge",
    "Intensity_MaxIntensityEdge",
]
INTENSITY_TAIL = [
    "Intensity_MassDisplacement",
    "Intensity_LowerQuartileIntensity",
    "Intensity_MedianIntensity",
    "Intensity_MADIntensity",
    "Intensity_UpperQuartileIntensity",
    "Location_CenterMassIntensity_X",
    "Location_CenterMassIntensity_Y",
    "Location_CenterMassIntensity_Z",
    "Location_MaxIntensity_X",
    "Location_MaxIntensity_Y",
    "Location_MaxIntensity_Z",
]


def intensity_names(edge_measurements: bool = True) -> list[str]:
    """Column order written by k_intensity (feat_intensity.hip)."""
    return INTENSITY_CORE + (INTENSITY_EDGE if edge_measurements else []) + INTENSITY_TAIL


SIZESHAPE_BASIC = [
    "Area",
    "BoundingBoxArea",
    "BoundingBoxMaximum_X",
    "BoundingBoxMaximum_Y",
    "BoundingBoxMinimum_X",
    "BoundingBoxMinimum_Y",
    "Center_X",
    "Center_Y",
    "Compactness",
    "ConvexArea",
    "Eccentricity",
    "EquivalentDiameter",
    "EulerNumber",
    "Extent",
    "FormFactor",
    "MajorAxisLength",
    "MaxFeretDiameter",
    "MaximumRadius",
    "MeanRadius",
    "MedianRadius",
    "MinFeretDiameter",
    "MinorAxisLength",
    "Orientation",
    "Perimeter",
    "Solidity",
]


def sizeshape_names() -> list[str]:
    """Column order written by k_shape_* (feat_shape.hip): 25 basic + 53 advanced = 78."""
    names = list(SIZESHAPE_BASIC)
    names += [f"SpatialMoment_{p}_{q}" for p in range(3) for q in range(4)]
    names += [f"CentralMoment_{p}_{q}" for p in range(3) for q in range(4)]
    names += [f"NormalizedMoment_{p}_{q}" for p in range(4) for q in range(4)]
    names += [f"HuMoment_{k}" for k in range(7)]
    names += [f"InertiaTensor_{i}_{j}" for i in range(2) for j in range(2)]
    names += [f"InertiaTensorEigenvalues_{k}" for k in range(2)]
    assert len(names) == 78
    return names


def zernike_indexes(limit: int = 10) -> list[tuple[int, int]]:
    """(n, m) pairs with n < limit and m = n%2, n%2+2, .., n  (centrosome.zernike.get_zernike_indexes)."""
    return [(n, m) for n in range(limit) for m in range(n % 2, n + 1, 2)]


def zernike_names() -> list[str]:
    return [f"Zernike_{n}_{m}" for n, m in zernike_indexes()]


def feret_names() -> list[str]:
    return ["MinFeretDiameter", "MaxFeretDiameter"]


HARALICK = [
    "AngularSecondMoment",
    "Contrast",
    "Correlation",
    "Variance",
    "InverseDifferenceMoment",
    "SumAverage",
    "SumVariance",
    "SumEntropy",
    "Entropy",
    "DifferenceVariance",
    "DifferenceEntropy",
    "InfoMeas1",
    "InfoMeas2",
]


def texture_names(scale: int = 3, gray_levels: int = 256) -> list[str]:
    """13 Haralick statistics x 4 directions; direction-major like mahotas' (4, 13) result."""
    return [f"{h}_{scale}_{d:02d}_{gray_levels}" for d in range(4) for h in HARALICK]


def radial_distribution_names(bin_count: int = 4, scaled: bool = True) -> list[str]:
    """scaled=False adds CellProfiler's overflow ring (everything beyond maximum_radius) as `<stat>_Overflow`."""
    out = []
    for stat in ("FracAtD", "MeanFrac", "RadialCV"):
        out += [f"RadialDistribution_{stat}_{b}of{bin_count}" for b in range(1, bin_count + 1)]
        if not scaled:
            out.append(f"RadialDistribution_{stat}_Overflow")
    return out


def granularity_names(granular_spectrum_length: int = 16) -> list[str]:
    return [f"Granularity_{i}" for i in range(1, int(granular_spectrum_length) + 1)]


def radial_zernike_names() -> list[str]:
    out = [f"RadialDistribution_ZernikeMagnitude_{n}_{m}" for n, m in zernike_indexes()]
    out += [f"RadialDistribution_ZernikePhase_{n}_{m}" for n, m in zernike_indexes()]
    return out


COLOC = {
    "pearson": ["Correlation_Pearson", "Correlation_Slope"],
    "manders_fold": ["Correlation_Manders_1", "Correlation_Manders_2"],
    "rwc": ["Correlation_RWC_1", "Correlation_RWC_2"],
    "costes": ["Correlation_Costes_1", "Correlation_Costes_2"],
}


def family_names(family: str, **kw) -> list[str]:
    if family == "intensity":
        return intensity_names(kw.get("edge_measurements", True))
    if family == "sizeshape":
        return sizeshape_names()
    if family == "zernike":
        return zernike_names()
    if family == "feret":
        return feret_names()
    if family == "texture":
        return texture_names(kw.get("scale", 3), kw.get("gray_levels", 256))
    if family == "radial_distribution":
        return radial_distribution_names(kw.get("bin_count", 4), kw.get("scaled", True))
    if family == "radial_zernikes":
        return radial_zernike_names()
    if family in COLOC:
        return list(COLOC[family])
    raise KeyError(family)


def intensity3d_names() -> list[str]:
    """Round-3 extension (a Z-stack measured as a volume; beyond what the reference wires): the moment-based
    statistics of CellProfiler's MeasureObjectIntensity on volumes, named as cp_measure's 2-D `intensity` names them."""
    return ["Volume", "Intensity_IntegratedIntensity", "Intensity_MeanIntensity", "Intensity_StdIntensity", "Intensity_MinIntensity",
            "Intensity_MaxIntensity", "Location_CenterMassIntensity_X", "Location_CenterMassIntensity_Y", "Location_CenterMassIntensity_Z",
            "Location_Center_X", "Location_Center_Y", "Location_Center_Z"]
