"""Option/parameter record for the residual + smoother hot path.

Host-side mirror of the Fortran module variables the reference's hot path reads
(`src/modules/inputParam.F90:1-97` inputDiscretization, `:183-299`
inputIteration, `:507-635` inputPhysics, `src/modules/flowVarRefState.F90`,
`src/modules/iteration.f90`).  Names and enumeration values are the
reference's (`src/modules/constants.F90`) so a Fortran caller fills the same
record from its module variables.
"""
from __future__ import annotations

import dataclasses
import math
from dataclasses import dataclass, field
from typing import List

# enumerations: src/modules/constants.F90
EulerEquations, NSEquations, RANSEquations = 1, 2, 3
dissScalar, dissMatrix, upwind = 1, 2, 9
noLimiter, vanAlbeda, minmod = 2, 3, 4
RungeKutta, DADI = 1, 2
spalartAllmaras = 2
strain, vorticity, katoLaunder = 1, 2, 3
firstOrder, secondOrder = 1, 2
noResAveraging, alwaysResAveraging, alternateResAveraging = 0, 1, 2
noFlux, boundFlux, normalFlux = -1, 0, 1


@dataclass
class FlowParams:
    # --- inputPhysics
    equations: int = EulerEquations
    turbModel: int = spalartAllmaras
    turbProd: int = strain
    useQCR: bool = False
    useRotationSA: bool = False
    useft2SA: bool = True
    gammaConstant: float = 1.4
    prandtl: float = 0.72
    prandtlTurb: float = 0.90
    SSuthDim: float = 110.55
    muSuthDim: float = 1.716e-5
    TSuthDim: float = 273.15
    eddyVisInfRatio: float = 0.009
    SAKappa: float = 0.41
    SAcb1: float = 0.1355
    SAcb2: float = 0.622
    SAsigma: float = 0.66666666667
    SAcv1: float = 7.1
    SAcw2: float = 0.3
    SAcw3: float = 2.0
    SAct1: float = 1.0
    SAct2: float = 2.0
    SAct3: float = 1.2
    SAct4: float = 0.5
    SAcrot: float = 2.0
    # --- inputDiscretization
    spaceDiscr: int = dissScalar
    spaceDiscrCoarse: int = dissScalar
    limiter: int = vanAlbeda
    orderTurb: int = firstOrder
    vis2: float = 0.25
    vis4: float = 0.0156
    vis2Coarse: float = 0.5
    adis: float = 0.67
    acousticScaleFactor: float = 1.0
    kappaCoef: float = 1.0 / 3.0
    sigma: float = 0.0
    dirScaling: bool = True
    # --- inputIteration
    smoother: int = RungeKutta
    nRKStages: int = 5
    etaRK: List[float] = field(default_factory=lambda: [0.25, 1.0 / 6.0, 0.375, 0.5, 1.0])
    cdisRK: List[float] = field(default_factory=lambda: [1.0, 0.0, 0.56, 0.0, 0.44])
    cfl: float = 1.7
    cflCoarse: float = 1.0
    cflLimit: float = 1.5
    fcoll: float = 0.8
    smoop: float = 1.5
    resAveraging: int = noResAveraging
    turbResScale: float = 10000.0
    nSubiterations: int = 1
    nSubIterTurb: int = 3
    turbRelax: int = 2          # turbRelaxImplicit (SA default)
    # --- boundary treatment (inputDiscretization; pyADflow defaults: linear / constant / constant)
    eulerWallBCTreatment: int = 2
    viscWallBCTreatment: int = 1
    outflowTreatment: int = 1
    lowSpeedPreconditioner: bool = False
    hScalingInlet: bool = False
    exchangePressureEarly: bool = False   # iteration.f90:44: normal-momentum Euler walls present on any process
    LRef: float = 1.0
    ordersConverged: float = 16.0
    alfaTurb: float = 0.8
    # bit mask of reference features outside the GPU path (include/adflow_gpu.h adflow_opts::unsupported); 0 = none
    unsupported: int = 0
    betaTurb: float = -1.0
    # --- iteration
    currentLevel: int = 1
    groundLevel: int = 1
    rkStage: int = 0
    rFil: float = 1.0
    # --- flowVarRefState (non-dimensional reference state, referenceState
    #     initializeFlow.F90:10-182: pRef=pInfDim, rhoRef=rhoInfDim, TRef=TInfDim)
    Mach: float = 0.8
    alphaDeg: float = 1.8
    pInfDim: float = 26500.0
    rhoInfDim: float = 0.4135
    TInfDim: float = 223.25
    RGasDim: float = 287.055

    # ---- derived, as the reference derives them -----------------------------
    @property
    def viscous(self) -> bool:
        return self.equations in (NSEquations, RANSEquations)

    @property
    def eddyModel(self) -> bool:
        return self.equations == RANSEquations

    @property
    def nw(self) -> int:
        return 6 if self.equations == RANSEquations else 5

    @property
    def nwf(self) -> int:
        return 5

    @property
    def gammaInf(self) -> float:
        return self.gammaConstant

    @property
    def pInf(self) -> float:
        return 1.0

    @property
    def pInfCorr(self) -> float:
        return 1.0  # SA carries no k: pInfCorr = pInf

    @property
    def rhoInf(self) -> float:
        return 1.0

    @property
    def uInf(self) -> float:
        return self.Mach * math.sqrt(self.gammaInf * self.pInf / self.rhoInf)

    @property
    def RGas(self) -> float:
        return self.RGasDim * self.rhoInfDim * self.TInfDim / self.pInfDim

    @property
    def muRef(self) -> float:
        return math.sqrt(self.pInfDim * self.rhoInfDim)

    @property
    def TRef(self) -> float:
        return self.TInfDim

    @property
    def muInfDim(self) -> float:
        return (self.muSuthDim * ((self.TSuthDim + self.SSuthDim) / (self.TInfDim + self.SSuthDim))
                * (self.TInfDim / self.TSuthDim) ** 1.5)

    @property
    def muInf(self) -> float:
        return self.muInfDim / self.muRef

    @property
    def pRef(self) -> float:
        return self.pInfDim

    @property
    def uRef(self) -> float:
        return math.sqrt(self.pInfDim / self.rhoInfDim)

    @property
    def timeRef(self) -> float:
        return math.sqrt(self.rhoInfDim / self.pInfDim)

    @property
    def SAcw1(self) -> float:
        # initializeFlow / paramTurb: rsaCw1 = cb1/kappa^2 + (1+cb2)/cb3
        return self.SAcb1 / self.SAKappa ** 2 + (1.0 + self.SAcb2) / self.SAsigma

    @property
    def velDirFreestream(self):
        a = math.radians(self.alphaDeg)
        return (math.cos(a), math.sin(a), 0.0)

    def wInf(self):
        d = self.velDirFreestream
        w = [self.rhoInf, self.uInf * d[0], self.uInf * d[1], self.uInf * d[2], 0.0]
        # rhoE_inf: etot of the free stream
        v2 = self.uInf ** 2
        w[4] = self.pInf / (self.gammaInf - 1.0) + 0.5 * self.rhoInf * v2
        if self.nw > 5:
            w.append(sa_nu_known_eddy_ratio(self.eddyVisInfRatio, self.muInf / self.rhoInf, self.SAcv1))
        return w

    def replace(self, **kw) -> "FlowParams":
        return dataclasses.replace(self, **kw)


def sa_nu_known_eddy_ratio(eddy_ratio: float, nu_lam: float, cv1: float) -> float:
    """nuTilde for a prescribed eddy/laminar viscosity ratio: Newton solve of
    chi^4 - ratio*chi^3 - ratio*cv1^3 = 0 (turbUtils.F90 saNuKnownEddyRatio)."""
    if eddy_ratio <= 0.0:
        return 0.0
    cv13 = cv1 ** 3
    if eddy_ratio < 1e-4:
        chi = 0.5
    elif eddy_ratio < 1.0:
        chi = 5.0
    elif eddy_ratio < 10.0:
        chi = 10.0
    else:
        chi = eddy_ratio
    for _ in range(100):
        chi2 = chi * chi
        chi3 = chi2 * chi
        f = chi3 * chi - eddy_ratio * (chi3 + cv13)
        df = 4.0 * chi3 - 3.0 * eddy_ratio * chi2
        dchi = f / df
        chi -= dchi
        if abs(dchi / chi) <= 1e-12:
            break
    return nu_lam * chi
