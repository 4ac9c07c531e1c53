// v_SR(d): the real-space pair kernel and its derivative, shared by the pair kernels.
// Reference: Potential.sr_from_dist / from_dist / f_cutoff (potentials/potential.py:59-138),
// CoulombPotential (potentials/coulomb.py:80-120), InversePowerLawPotential (potentials/inversepowerlaw.py:55-106).
#pragma once

#include "common.h"

namespace mipme {

static constexpr double kPiR = 3.14159265358979323846;

// Device-side description of v_SR(d)
struct SRPot {
  int mode;        // 0: bare v, 1: v - v_LR (range separated), 2: -v_LR * f_cut, 3: v * (1 - f_cut)
  int p;           // exponent 1..6
  double pref;
  double inv_2s2;  // 1/(2 sigma^2)
  double rx;       // exclusion radius
  int deg;         // exclusion degree
};

inline int make_srpot(const mipme_potential_t* pot, SRPot& s) {
  MIPME_REQUIRE(pot != nullptr, "potential descriptor is NULL");
  s.p = pot->kind == MIPME_COULOMB ? 1 : pot->exponent;
  MIPME_REQUIRE(s.p >= 1 && s.p <= 6, "Unsupported exponent: %d", s.p);
  const bool smeared = pot->smearing > 0;
  const bool excl = pot->exclusion_radius > 0;
  s.mode = smeared ? (excl ? 2 : 1) : (excl ? 3 : 0);
  s.pref = pot->prefactor;
  s.inv_2s2 = smeared ? 0.5 / (pot->smearing * pot->smearing) : 0.0;
  s.rx = pot->exclusion_radius;
  s.deg = pot->exclusion_degree;
  return MIPME_OK;
}

__device__ __forceinline__ float fexp(float x) { return expf(x); }
__device__ __forceinline__ double fexp(double x) { return exp(x); }
__device__ __forceinline__ float ferfc(float x) { return erfcf(x); }
__device__ __forceinline__ double ferfc(double x) { return erfc(x); }
__device__ __forceinline__ float fsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }
__device__ __forceinline__ float fsin(float x) { return sinf(x); }
__device__ __forceinline__ double fsin(double x) { return sin(x); }
__device__ __forceinline__ float fcos(float x) { return cosf(x); }
__device__ __forceinline__ double fcos(double x) { return cos(x); }

// erfc(y) given e = exp(-y^2), y >= 0.  float: erfc = e * t * P8(t), t = 1/(1 + 0.4 y), P8 fitted (Chebyshev nodes) to
// erfcx(y)/t on [0, 6.5]: relative error <= 3.7e-7 in fp32 arithmetic (the libm erfcf costs ~4x the instructions
// and made the pair kernels VALU-bound); beyond y = 6.5, where erfc < 4e-20, the error grows to 2e-5 relative.
__device__ __forceinline__ float erfc_from_exp(float y, float e) {
  const float t = 1.0f / (1.0f + 0.4f * y);
  float p = 2.646481385e-02f;
  p = p * t + -6.557867191e-02f;
  p = p * t + -7.738398321e-02f;
  p = p * t + 2.820383187e-01f;
  p = p * t + -6.220284696e-02f;
  p = p * t + 2.579626189e-01f;
  p = p * t + 1.840778096e-01f;
  p = p * t + 2.291638826e-01f;
  p = p * t + 2.254580744e-01f;
  return e * t * p;
}
// double: erfc(y) = e * t * P20(x), x = (2 t - (1 + tlo)) / (1 - tlo) in [-1, 1] for t in [1/(1+0.4*27), 1]: a degree-20 fit of
// erfcx(y)/t at Chebyshev nodes (max relative error 2e-15 for 0 <= y <= 27, beyond which e = exp(-y^2) underflows anyway), given
// by its MONOMIAL coefficients in x and evaluated with Horner's rule: 20 FMAs (the Chebyshev form with Clenshaw's recurrence,
// round 2, needed 40 operations for the same accuracy; the coefficients decay faster than the basis grows, so the monomial
// form loses nothing: 3.8e-15 against 4.2e-15).
#define MIPME_ERFC_CHEB                                                                                                   \
  {4.50235299459692540e-01,  3.24125536248557999e-01,  1.67075628387939157e-01,  5.52651679935470402e-02,                \
   6.82018143828926407e-03,  -2.76262477755241610e-03, -1.03953621374069873e-03, 1.92114029446880925e-04,                \
   1.22950360871657786e-04,  -2.59317092799155009e-05, -1.43585248578882984e-05, 5.12388472713849280e-06,                \
   1.35756476325816326e-06,  -1.02658932393014442e-06, -3.47187889529154472e-09, 1.73400820276635575e-07,                \
   -4.17454076893041247e-08, -2.09125782862454419e-08, 1.11490864632721937e-08,  1.28815160249804723e-09,                \
   -1.26127363551349624e-09}
static constexpr int kErfcChebTerms = 21;
// ---- fp64 pair body: erfcx(y) = exp(y^2) erfc(y) from a table of local polynomials ---------------------------------------
// (tools/gen_erfcx_table.py 0.125 6.5 8)
// erfcx(y), y in [0, 6.5): 52 intervals of width 0.125, degree 8 in u = y - centre; max relative error 2.22e-15
// Rows of kErfcxRow doubles (9 coefficients, lowest first, + one pad: 16-byte aligned pairs for ds_read_b128), copied to LDS by
// every row workgroup; behind them the 21 coefficients of the whole-range fit (MIPME_ERFC_CHEB) for the rare y >= kErfcxEnd.
// erfc(y) = e * erfcx(y) then costs int(8 y), one fma for u, five 16-byte LDS reads and 8 FMAs -- the whole-range form needs a
// Newton reciprocal (5) + 21 FMAs and, with its 21 double constants next to exp's 14 in scalar registers, made the compiler
// spill SGPRs to VGPR lanes and re-load kernel arguments inside the pair loop.
static constexpr int kErfcxIntervals = 52, kErfcxTerms = 9, kErfcxRow = 10;
static constexpr double kErfcxWidthInv = 8.0, kErfcxWidth = 0.125, kErfcxEnd = 6.5;
static constexpr int kErfcxLdsDoubles = kErfcxIntervals * kErfcxRow + 22;  // + whole-range coefficients (21, padded)
#define MIPME_ERFCX_TAB \
  9.33206248649274372e-01, -1.01172838601424786e+00, 8.69973224522498700e-01, -6.38236706714343072e-01, 4.15041716280737105e-01, -2.44918265107563010e-01, 1.33244250186039193e-01, -6.77258910872031855e-02, 3.23541616175097996e-02, 0.0, \
  8.19181308058672508e-01, -8.21186176573459603e-01, 6.65208899950698207e-01, -4.64306338766932669e-01, 2.89075730817692889e-01, -1.64041644685596955e-01, 8.61059559821657772e-02, -4.23305955724228786e-02, 1.95617205634364549e-02, 0.0, \
  7.26085955123769722e-01, -6.74575445143114849e-01, 5.15281128516132059e-01, -3.42366728473263993e-01, 2.04145763483998877e-01, -1.11428330732147696e-01, 5.64411922845786951e-02, -2.68438793681257185e-02, 1.20650462744919286e-02, 0.0, \
  6.49193250053864856e-01, -5.60335073298365072e-01, 4.04046655486651973e-01, -2.55709774426437386e-01, 1.46086813487226852e-01, -7.67186378731629509e-02, 3.75079038903928588e-02, -1.72587111899998626e-02, 7.44223189112287742e-03, 0.0, \
  5.84998621474965952e-01, -4.70255717936147122e-01, 3.20479780136086523e-01, -1.93323894476729069e-01, 1.05867544550001413e-01, -5.35093038339467947e-02, 2.52562275786443677e-02, -1.12470396191271388e-02, 4.73620544700996318e-03, 0.0, \
  5.30870872417554596e-01, -3.98431717521372575e-01, 2.56949066621900535e-01, -1.47852822830826597e-01, 7.76501252772753420e-02, -3.77873199493784923e-02, 1.72238223555245158e-02, -7.42287875865051561e-03, 3.03039640034849858e-03, 0.0, \
  4.84810828561620388e-01, -3.40561570682877568e-01, 2.08104552381388891e-01, -1.14317747929925120e-01, 5.76106917433175797e-02, -2.70036079811914950e-02, 1.18898038749696774e-02, -4.96130100054810019e-03, 2.00314996604697253e-03, 0.0, \
  4.45282273136881956e-01, -2.93474904963846928e-01, 1.70149549732939914e-01, -8.93064680813515366e-02, 4.32123682419868621e-02, -1.95179344400502365e-02, 8.30467068207033088e-03, -3.35641961176115326e-03, 1.30242070812500968e-03, 0.0, \
  4.11092054444830657e-01, -2.54808551400252825e-01, 1.40357968582010079e-01, -7.04521398495527723e-02, 3.27512851138245312e-02, -1.42615579528181120e-02, 5.86606133357479419e-03, -2.29581408183612530e-03, 8.67852630019155021e-04, 0.0, \
  3.81304058966718151e-01, -2.22782027049563097e-01, 1.16750401844990401e-01, -5.60939499009696571e-02, 2.50694185262449326e-02, -1.05296058594290393e-02, 4.18837149833173761e-03, -1.58841967307135558e-03, 5.92401852743630803e-04, 0.0, \
  3.55176786497634234e-01, -1.96040102539221495e-01, 9.78741519144962629e-02, -4.50535187722897815e-02, 1.93707047781783055e-02, -7.85178348763612872e-03, 3.02153652117990263e-03, -1.11150445166973886e-03, 4.17920457447744625e-04, 0.0, \
  3.32117562728372284e-01, -1.73541174251443858e-01, 8.26521247422738747e-02, -3.64858299538123598e-02, 1.51018716974351420e-02, -5.91075547037431866e-03, 2.20187946392240994e-03, -7.85019168876333029e-04, 2.47278208098336333e-04, 0.0, \
  3.11648608648130210e-01, -1.54477265070108560e-01, 7.02778819759482948e-02, -2.97787163213675231e-02, 1.18743189885049949e-02, -4.49003558149417472e-03, 1.61951257670151839e-03, -5.60481465445319916e-04, 1.88666904333150098e-04, 0.0, \
  2.93381648765277336e-01, -1.38216102512704775e-01, 6.01419757750265918e-02, -2.44843455894424820e-02, 9.41232121174125601e-03, -3.44042332772659084e-03, 1.20225613004431285e-03, -4.03299686050222466e-04, 1.23130261575102963e-04, 0.0, \
  2.76998730673052806e-01, -1.24258768405697231e-01, 5.17797129379564541e-02, -2.02720258018643067e-02, 7.51833279152560371e-03, -2.65801942820916622e-03, 9.00342491911478721e-04, -2.93314160485745330e-04, 7.79518948524469973e-05, 0.0, \
  2.62237606550381475e-01, -1.12208441712784684e-01, 4.48337507320319892e-02, -1.68953664457908229e-02, 6.04948899028453016e-03, -2.06979297199677689e-03, 6.79786925625967082e-04, -2.15066357131813032e-04, 6.39496397263310793e-05, 0.0, \
  2.48880496184162359e-01, -1.01747120335846791e-01, 3.90270604915579772e-02, -1.41692053708548111e-02, 4.90153721705966026e-03, -1.62391971650973363e-03, 5.17383388319524054e-04, -1.58272325652230163e-04, 5.04603206733031457e-05, 0.0, \
  2.36745378740146423e-01, -9.26181351073768339e-02, 3.41432081925044095e-02, -1.19532447832970349e-02, 3.99774290164255086e-03, -1.28327704797231166e-03, 3.96744233088841829e-04, -1.18026570585582754e-04, 4.95974842117334055e-05, 0.0, \
  2.25679191606819457e-01, -8.46129059139759665e-02, 3.00118466808273268e-02, -1.01403403027719337e-02, 3.28115478940806593e-03, -1.02107208926707077e-03, 3.06663859719851341e-04, -8.84450661298663918e-05, 2.40121506459268445e-05, 0.0, \
  2.15552479151177590e-01, -7.75608312335239053e-02, 2.64979530194129030e-02, -8.64804715925920550e-03, 2.70916908074093346e-03, -8.17783353502546991e-04, 2.38586216412022602e-04, -6.67542882089119683e-05, 2.24765341402595187e-05, 0.0, \
  2.06255152386500912e-01, -7.13215111146973529e-02, 2.34937801551998206e-02, -7.41246630622863743e-03, 2.24966752933100861e-03, -6.59080375395947331e-04, 1.86950998871782505e-04, -5.09605696483749116e-05, 1.19771882762257653e-05, 0.0, \
  1.97693106149973097e-01, -6.57787215394098346e-02, 2.09127920127080320e-02, -6.38372866434668614e-03, 1.87826077324827778e-03, -5.34364815309054599e-04, 1.47307761330156691e-04, -3.89112132346302033e-05, 2.21849205821911418e-05, 0.0, \
  1.89785502908994708e-01, -6.08357132324193264e-02, 1.86850594426259953e-02, -5.52265569561000667e-03, 1.57629535453678959e-03, -4.35733110832905838e-04, 1.16857800914642071e-04, -3.00085459433406328e-05, 1.70504631990801298e-05, 0.0, \
  1.82462578344034759e-01, -5.64115193243117236e-02, 1.67537403288393963e-02, -4.79827140069358140e-03, 1.32940909546995739e-03, -3.57255245284494492e-04, 9.33022883289244114e-05, -2.34130777093760782e-05, 8.83464904326395347e-06, 0.0, \
  1.75663858002584439e-01, -5.24380368296841065e-02, 1.50723702115042150e-02, -4.18593536905190709e-03, 1.12647175510275613e-03, -2.94447809360609547e-04, 7.48385131082931088e-05, -1.83386520182850288e-05, 1.37218926191945231e-05, 0.0, \
  1.69336699837247806e-01, -4.88577056330586207e-02, 1.36027631319298181e-02, -3.66593209745147853e-03, 9.58802233380663888e-04, -2.43901760324731622e-04, 6.04731361205680052e-05, -1.43164233431739640e-05, 1.69477334300960263e-06, 0.0, \
  1.63435096646622369e-01, -4.56216518116421346e-02, 1.23133750205740890e-02, -3.22239803268876491e-03, 8.19590755143131218e-04, -2.03004001264149304e-04, 4.90520452146507192e-05, -1.11761174086244548e-05, 2.42134225161442845e-06, 0.0, \
  1.57918686990727669e-01, -4.26881940342603017e-02, 1.11780199978541769e-02, -2.84250019220145698e-03, 7.03462933584187680e-04, -1.69740330995335698e-04, 3.99319806890751687e-05, -8.92914356278510217e-06, 1.09074522119763162e-05, 0.0, \
  1.52751934252847554e-01, -4.00216355439728458e-02, 1.01748576273945120e-02, -2.51580349729179243e-03, 6.06153920071771709e-04, -1.42553176528381338e-04, 3.27270053836669825e-05, -7.13312063091858439e-06, 8.08095842990278305e-06, 0.0, \
  1.47903442039590050e-01, -3.75912820535371367e-02, 9.28558946720665565e-03, -2.23378059333231588e-03, 5.24261703189828781e-04, -1.20227656140120846e-04, 2.69985443191586055e-05, -5.65603693199745664e-06, -9.40329121379989136e-07, 0.0, \
  1.43345380690332175e-01, -3.53706393317317208e-02, 8.49481823801164285e-03, -1.98942986171686542e-03, 4.55058547058472083e-04, -1.01810169248227916e-04, 2.22639480263965553e-05, -4.42628405601213211e-06, 6.18365104427611374e-06, 0.0, \
  1.39053004777814626e-01, -3.33367544702247581e-02, 7.78953405123417302e-03, -1.77697609186514135e-03, 3.96345421383037825e-04, -8.65485997323188862e-05, 1.84900837703998618e-05, -3.54878877459515494e-06, 5.25013127851729074e-06, 0.0, \
  1.35004245473810763e-01, -3.14696726208002209e-02, 7.15870045165095670e-03, -1.59163469172616070e-03, 3.46342420699421289e-04, -7.38466363737508524e-05, 1.53862386593719295e-05, -3.34671257250790588e-06, 7.99880728755814075e-06, 0.0, \
  1.31179364789273029e-01, -2.97519869853511247e-02, 6.59291928803888271e-03, -1.42942497843772721e-03, 3.03601149444377414e-04, -6.32376864999279809e-05, 1.29154652503273937e-05, -2.65897034270634423e-06, 2.23157865812689552e-06, 0.0, \
  1.27560661173924755e-01, -2.81684644704124909e-02, 6.08415814530074376e-03, -1.28702164583570692e-03, 2.66938643904586481e-04, -5.43393680470139091e-05, 1.08587108322541626e-05, -2.14786164837537664e-06, 2.62145244717134735e-06, 0.0, \
  1.24132217924707577e-01, -2.67057330137317611e-02, 5.62552767631605780e-03, -1.16163596953631001e-03, 2.35384011003429038e-04, -4.68462801247415504e-05, 9.16840272337718461e-06, -1.97059422103696233e-06, 7.76006441556428680e-07, 0.0, \
  1.20879687418954523e-01, -2.53520193975535282e-02, 5.21109891762338521e-03, -1.05092038957326070e-03, 2.08137312946995876e-04, -4.05179096483494484e-05, 7.75822205940532070e-06, -1.42052140128247153e-06, 7.00552896219747484e-07, 0.0, \
  1.17790105443152979e-01, -2.40969285659529131e-02, 4.83575279033391769e-03, -9.52891576808305527e-04, 1.84536678749606389e-04, -3.51486483164506982e-05, 6.61967213216069108e-06, -1.46770189042500113e-06, -2.87786473161761669e-06, 0.0, \
  1.14851730898194934e-01, -2.29312572003856179e-02, 4.49505562129655020e-03, -8.65868017336900468e-04, 1.64032949682233099e-04, -3.05826781456889225e-05, 5.59282637130516998e-06, -1.19278786411589090e-06, 3.52517671979467242e-06, 0.0, \
  1.12053906978460849e-01, -2.18468356832110452e-02, 4.18515579261743466e-03, -7.88419306037427335e-04, 1.46167747016657731e-04, -2.66858493392981953e-05, 4.79236535485024443e-06, -9.29819558664077413e-07, 1.71923100297508443e-06, 0.0, \
  1.09386940584858816e-01, -2.08363936738173196e-02, 3.90269761105724311e-03, -7.19324679055057328e-04, 1.30558352141837384e-04, -2.33486933092330399e-05, 4.05929812116072927e-06, -8.65057946442502934e-07, 7.66904564804224510e-06, 0.0, \
  1.06841997272159289e-01, -1.98934453968599885e-02, 3.64474927592246898e-03, -6.57539020049352865e-04, 1.16882840534561749e-04, -2.04826825542592281e-05, 3.52778944248617357e-06, -7.93914999379104123e-07, 1.83605285975363536e-06, 0.0, \
  1.04411009473045283e-01, -1.90121914444067730e-02, 3.40874242462852134e-03, -6.02164876890162238e-04, 1.04870756933311864e-04, -1.80147469842354640e-05, 3.06041644389900138e-06, -6.46107983456051322e-07, -1.10028483632840581e-06, 0.0, \
  1.02086596104440236e-01, -1.81874344597257841e-02, 3.19242122961920904e-03, -5.52429349742906734e-04, 9.42933841732241165e-05, -1.58830048530666623e-05, 2.62011860307090973e-06, -5.71329072038735707e-07, 2.55045053762188864e-06, 0.0, \
  9.98619919610877066e-02, -1.74145065284111548e-02, 2.99379939680327872e-03, -5.07664924461348087e-04, 8.49565954313665433e-05, -1.40363206619099760e-05, 2.31194628203987515e-06, -5.51401596342128565e-07, -2.98551062003821858e-06, 0.0, \
  9.77309855491024781e-02, -1.66892064744720327e-02, 2.81112372561099491e-03, -4.67293524490353573e-04, 7.66958415509903351e-05, -1.24335795209706092e-05, 2.01196937460155649e-06, -4.29154366173979995e-07, -1.94180407724252549e-06, 0.0, \
  9.56878642179170430e-02, -1.60077455622280283e-02, 2.64284313746636183e-03, -4.30813216096530098e-04, 6.93706908652212407e-05, -1.10389652674586692e-05, 1.71714979273461307e-06, -1.98363452058357089e-07, 2.95803798308240924e-06, 0.0, \
  9.37273656204213046e-02, -1.53667003530101858e-02, 2.48758227444825872e-03, -3.97787065521403222e-04, 6.28607750161789370e-05, -9.82027029794074354e-06, 1.52069137714977476e-06, -2.90343146826991798e-07, -4.96481004158642758e-07, 0.0, \
  9.18446346743245073e-02, -1.47629716693277666e-02, 2.34411892907887640e-03, -3.67833775570390977e-04, 5.70632988964194413e-05, -8.75438733773692174e-06, 1.33555152389127134e-06, -2.85846655702570126e-07, -8.40494236880150709e-08, 0.0, \
  9.00351853178581879e-02, -1.41937487870182535e-02, 2.21136469819802322e-03, -3.40619810588638488e-04, 5.18897991221317254e-05, -7.82099617239340301e-06, 1.16655987425304570e-06, -1.20225204846622247e-07, 4.93513555486588514e-07, 0.0, \
  8.82948664539331407e-02, -1.36564781146061801e-02, 2.08834835546272204e-03, -3.15852748757964902e-04, 4.72639628903262949e-05, -6.99870478708957639e-06, 1.01787242400294169e-06, -2.93722229923738502e-07, 1.18324992016220785e-06, 0.0, \
  8.66198315620469173e-02, -1.31488357341585250e-02, 1.97420152342907881e-03, -2.93275618267059470e-04, 4.31198363775559359e-05, -6.27664759428666796e-06, 9.14139800428344370e-07, -1.06324314045019544e-07, -8.88781796366163342e-07, 0.0

// 1/x and 1/sqrt(x) in double from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-26) and two Newton steps each: within an
// ulp or two of the IEEE sequences (v_div_scale / fmas / fixup, sqrt + division) at a third of their instruction count
__device__ __forceinline__ double rcp_newton(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double rsqrt_newton(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  y = y * __builtin_fma(-0.5 * x * y, y, 1.5);
  return y;
}
// c: the 21 coefficients.  The fused pair kernels pass the copy that travels in their kernel arguments (FastRS::cheb): it is
// loaded once into scalar registers, where 21 double literals would be re-materialised with two v_mov each per evaluation
// (65 of the ~300 instructions of the fp64 pair loop).
__device__ __forceinline__ double erfc_from_exp(double y, double e, const double* __restrict__ c) {
  constexpr double tlo = 0.0847457627118644;
  constexpr double xs = 2.0 / (1.0 - tlo), x0 = -(1.0 + tlo) / (1.0 - tlo);
  const double t = rcp_newton(1.0 + 0.4 * y);
  const double x = __builtin_fma(xs, t, x0);  // (2 t - (1 + tlo)) / (1 - tlo)
  double p = c[kErfcChebTerms - 1];
#pragma unroll
  for (int k = kErfcChebTerms - 2; k >= 0; --k) p = __builtin_fma(p, x, c[k]);
  return e * t * p;
}
// exp(-x), x >= 0, in double: n = round(-x log2 e), r = -x - n ln 2 (two-step reduction), e^r from its degree-13 Taylor
// polynomial (|r| <= ln 2 / 2: 2e-16), scaled by 2^n -- 18 instructions where libm's exp takes about twice as many (its range
// checks and table look-ups are for arguments this one never sees)
__device__ __forceinline__ double exp_neg_fast(double x) {
  x = __builtin_fmin(x, 700.0);
  const double n = __builtin_rint(-x * 1.4426950408889634);
  double r = __builtin_fma(n, -0.693147180369123816490, -x);
  r = __builtin_fma(n, -1.90821492927058770002e-10, r);
  double p = 1.6059043836821613e-10;  // 1/13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);
  p = __builtin_fma(p, r, 2.505210838544172e-08);
  p = __builtin_fma(p, r, 2.755731922398589e-07);
  p = __builtin_fma(p, r, 2.7557319223985893e-06);
  p = __builtin_fma(p, r, 2.48015873015873e-05);
  p = __builtin_fma(p, r, 0.0001984126984126984);
  p = __builtin_fma(p, r, 0.001388888888888889);
  p = __builtin_fma(p, r, 0.008333333333333333);
  p = __builtin_fma(p, r, 0.041666666666666664);
  p = __builtin_fma(p, r, 0.16666666666666666);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return __builtin_amdgcn_ldexp(p, int(n));
}
// the table in device memory (one copy per translation unit that uses it) and its staging into a workgroup's LDS
static __device__ const double kErfcxTabDevice[kErfcxIntervals * kErfcxRow] = {MIPME_ERFCX_TAB};
static __device__ const double kErfcChebDevice[kErfcChebTerms + 1] = MIPME_ERFC_CHEB;
__device__ __forceinline__ void erfcx_table_to_lds(double* __restrict__ lds, int tid, int nthr) {
  for (int k = tid; k < kErfcxIntervals * kErfcxRow; k += nthr) lds[k] = kErfcxTabDevice[k];
  for (int k = tid; k < kErfcChebTerms; k += nthr) lds[kErfcxIntervals * kErfcxRow + k] = kErfcChebDevice[k];
}
// exp(-x) with a table (round 5; the fp64 pair body is bound by fp64 issue at ~8 clocks per wavefront and instruction):
// -x = (64 m + j) ln2 / 64 + r, |r| <= ln2 / 128, exp(-x) = 2^m T[j] exp(r) with T[j] = 2^(j/64) from LDS and a degree-5 Taylor
// polynomial for exp(r) (r^6 / 720 < 3.5e-17) -- 5 dependent FMAs where exp_neg_fast2 below has 13; max relative error 4e-16
// (the 13-term form: 3e-16).  kExp2Tab doubles behind the erfcx table in the body's LDS (exp2_table_to_lds).
static constexpr int kExp2Tab = 64;
#define MIPME_EXP2_TAB                                                                                                        \
  1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284, 1.0442737824274138, 1.0556451783605572, 1.0671404006768237, \
  1.0787607977571199, 1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418, 1.1387886347566916,       \
  1.1511892299529827, 1.1637248587775775, 1.1763969916502812, 1.189207115002721, 1.202156731452703, 1.215247359980469,         \
  1.22848053610687, 1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783, 1.2968395546510096,          \
  1.3109612115247644, 1.3252366431597413, 1.339667524053303, 1.3542555469368927, 1.3690024229745905, 1.383909881963832,        \
  1.3989796725383112, 1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647, 1.4768261459394993,         \
  1.4929077282912648, 1.5091644275934228, 1.5255981507445384, 1.5422108254079407, 1.559004400237837, 1.5759808451078865,       \
  1.593142151342267, 1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364, 1.681792830507429,         \
  1.7001063537185235, 1.718619298122478, 1.7373338352737062, 1.7562521603732995, 1.7753764925265212, 1.7947090750031072,       \
  1.8142521755003989, 1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656, 1.9152065613971474,         \
  1.9360617934922943, 1.9571441241754002, 1.978456026387951
static __device__ const double kExp2TabDevice[kExp2Tab] = {MIPME_EXP2_TAB};
__device__ __forceinline__ void exp2_table_to_lds(double* __restrict__ lds, int tid, int nthr) {
  for (int k = tid; k < kExp2Tab; k += nthr) lds[k] = kExp2TabDevice[k];
}
__device__ __forceinline__ void exp_neg_table2(const double (&xin)[2], double (&out)[2], const double* __restrict__ tab) {
  double nf[2], r[2], q[2];
  int n[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const double x = __builtin_fmin(xin[u], 700.0);
    nf[u] = __builtin_rint(-x * 92.33248261689366);                 // 64 / ln2
    r[u] = __builtin_fma(nf[u], -0.01083042469326756, -x);          // ln2 / 64: 32 significant bits (n * hi is exact)
    r[u] = __builtin_fma(nf[u], -2.9815858269852933e-12, r[u]);     // ... and the rest
    n[u] = int(nf[u]);
    q[u] = 1.0 / 120.0;
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0 / 24.0);
#pragma unroll
  for (int u = 0; u < 2; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0 / 6.0);
#pragma unroll
  for (int u = 0; u < 2; ++u) q[u] = __builtin_fma(q[u], r[u], 0.5);
#pragma unroll
  for (int u = 0; u < 2; ++u) q[u] = __builtin_fma(q[u], r[u], 1.0);
#pragma unroll
  for (int u = 0; u < 2; ++u)
    out[u] = __builtin_amdgcn_ldexp(tab[n[u] & (kExp2Tab - 1)] * __builtin_fma(q[u], r[u], 1.0), n[u] >> 6);
}
// erfc(y) given e = exp(-y^2), y >= 0, with the table in LDS
__device__ __forceinline__ double erfc_from_table(double y, double e, const double* __restrict__ lds) {
  struct alignas(16) D2 {
    double a, b;
  };
  int j = int(y * kErfcxWidthInv);
  j = j < kErfcxIntervals ? j : kErfcxIntervals - 1;
  const double u = __builtin_fma(double(j), -kErfcxWidth, y - 0.5 * kErfcxWidth);
  const D2* row = reinterpret_cast<const D2*>(lds + j * kErfcxRow);
  const D2 c01 = row[0], c23 = row[1], c45 = row[2], c67 = row[3], c8 = row[4];
  double p = c8.a;
  p = __builtin_fma(p, u, c67.b);
  p = __builtin_fma(p, u, c67.a);
  p = __builtin_fma(p, u, c45.b);
  p = __builtin_fma(p, u, c45.a);
  p = __builtin_fma(p, u, c23.b);
  p = __builtin_fma(p, u, c23.a);
  p = __builtin_fma(p, u, c01.b);
  p = __builtin_fma(p, u, c01.a);
  double r = e * p;
  if (y >= kErfcxEnd) {  // erfc < 4e-20: rare; the whole-range fit, coefficients read from LDS
    const double* c = lds + kErfcxIntervals * kErfcxRow;
    constexpr double tlo = 0.0847457627118644;
    constexpr double xs = 2.0 / (1.0 - tlo), x0 = -(1.0 + tlo) / (1.0 - tlo);
    const double t = rcp_newton(1.0 + 0.4 * y);
    const double x = __builtin_fma(xs, t, x0);
    double q = c[kErfcChebTerms - 1];
    for (int k = kErfcChebTerms - 2; k >= 0; --k) q = __builtin_fma(q, x, c[k]);
    r = e * t * q;
  }
  return r;
}
// two at once, term by term (see sr_rows_f64_body)
__device__ __forceinline__ void exp_neg_fast2(const double (&xin)[2], double (&out)[2]) {
  constexpr double c[13] = {1.6059043836821613e-10, 2.08767569878681e-09,  2.505210838544172e-08, 2.755731922398589e-07,
                            2.7557319223985893e-06, 2.48015873015873e-05,  0.0001984126984126984, 0.001388888888888889,
                            0.008333333333333333,   0.041666666666666664, 0.16666666666666666,   0.5,
                            1.0};
  double n[2], r[2], p[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const double x = __builtin_fmin(xin[u], 700.0);
    n[u] = __builtin_rint(-x * 1.4426950408889634);
    r[u] = __builtin_fma(n[u], -0.693147180369123816490, -x);
    r[u] = __builtin_fma(n[u], -1.90821492927058770002e-10, r[u]);
    p[u] = c[0];
  }
#pragma unroll
  for (int k = 1; k < 13; ++k) {
#pragma unroll
    for (int u = 0; u < 2; ++u) p[u] = __builtin_fma(p[u], r[u], c[k]);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) out[u] = __builtin_amdgcn_ldexp(__builtin_fma(p[u], r[u], 1.0), int(n[u]));
}
__device__ __forceinline__ double erfc_from_exp(double y, double e) {
  constexpr double c[kErfcChebTerms] = MIPME_ERFC_CHEB;
  return erfc_from_exp(y, e, c);
}
// same fit with the hardware reciprocal (1 ulp) for t
__device__ __forceinline__ float erfc_from_exp_fast(float y, float e) {
  const float t = __builtin_amdgcn_rcpf(1.0f + 0.4f * y);
  float p = 2.646481385e-02f;
  p = p * t + -6.557867191e-02f;
  p = p * t + -7.738398321e-02f;
  p = p * t + 2.820383187e-01f;
  p = p * t + -6.220284696e-02f;
  p = p * t + 2.579626189e-01f;
  p = p * t + 1.840778096e-01f;
  p = p * t + 2.291638826e-01f;
  p = p * t + 2.254580744e-01f;
  return e * t * p;
}

template <typename T>
__device__ __forceinline__ T powi(T x, int n) {
  T r = T(1);
  for (int i = 0; i < n; ++i) r *= x;
  return r;
}

// Q(p/2, x) regularised upper incomplete gamma and x^(p/2-1) e^-x / Gamma(p/2) for integer p in 1..6.
//   integer a:      Q(a,x) = e^-x sum_{k<a} x^k/k!
//   half-integer a: Q(1/2,x) = erfc(sqrt x);  Q(a+1,x) = Q(a,x) + x^a e^-x / Gamma(a+1)
template <typename T>
__device__ __forceinline__ void upper_gamma(int p, T x, T& Q, T& dens) {
  const T ex = fexp(-x);
  if ((p & 1) == 0) {
    const int a = p / 2;  // 1,2,3
    T term = T(1), sum = T(1);
    for (int k = 1; k < a; ++k) {
      term *= x / T(k);
      sum += term;
    }
    Q = ex * sum;
    dens = ex * term;  // x^(a-1)/(a-1)!
  } else {
    const T sx = fsqrt(x);
    const T isp = T(0.56418958354775628695);  // 1/sqrt(pi) = 1/Gamma(1/2)
    Q = erfc_from_exp(sx, ex);
    // term_k = x^(k-1/2) e^-x / Gamma(k+1/2)
    T term = (x > T(0)) ? ex * isp / sx : T(0);  // k = 0: x^(-1/2)/Gamma(1/2)
    dens = term;
    for (int k = 1; k <= (p - 1) / 2; ++k) {
      term *= x / (T(k) - T(0.5));
      Q += term;
      dens = term;
    }
  }
}

// P(p/2, x) = 1 - Q, the regularised LOWER incomplete gamma, for small x from its power series
//   P(a, x) = x^a e^-x sum_{k>=0} x^k / Gamma(a + k + 1)
// (what the reference evaluates directly: torch.special.gammainc, inversepowerlaw.py:98-103).  1 - Q loses all relative
// precision as x -> 0 -- P ~ x^a / Gamma(a+1) -- which matters inside an exclusion radius (mode 2: v = -v_LR f_cut): in fp32
// with p = 5, 6 it rounded to zero below d ~ 0.13 sigma.  Used for x < 1, where 12 (float) / 22 (double) terms reach the
// rounding level (term ratio x / (a + k)).
template <typename T>
__device__ __forceinline__ T lower_gamma_series(int p, T x) {
  // 1 / Gamma(p/2 + 1), p = 1..6
  constexpr double inv_gamma[6] = {1.1283791670955126, 1.0, 0.7522527780636751, 0.5, 0.30090111122547003, 1.0 / 6.0};
  const T a = T(0.5) * T(p);
  T term = T(inv_gamma[p - 1]), sum = term;
  constexpr int K = sizeof(T) == 4 ? 12 : 22;
#pragma unroll
  for (int k = 1; k <= K; ++k) {
    term *= x / (a + T(k));
    sum += term;
  }
  T xa = powi(x, p / 2);
  if (p & 1) xa *= fsqrt(x);
  return xa * fexp(-x) * sum;
}

// v_SR(d) and dv_SR/dd (see oracle/pme_numpy.py::sr_pair for the derivation).
template <typename T, bool DERIV>
__device__ __forceinline__ void sr_eval(const SRPot& s, T d, T& v, T& dv) {
  const T pref = T(s.pref);
  const T dc = d < T(1e-15) ? T(1e-15) : d;  // (this way round a NaN distance stays NaN: the reference's sr_from_dist propagates it)
  const T inv = T(1) / dc;
  const T invp = powi(inv, s.p);
  T fc = T(0), dfc = T(0);
  if (s.mode >= 2) {
    const T rx = T(s.rx);
    if (d < rx) {
      const T arg = T(kPiR) * d / rx;
      const T base = T(0.5) * (T(1) - fcos(arg));
      fc = T(1) - powi(base, s.deg);
      if constexpr (DERIV) dfc = -T(s.deg) * powi(base, s.deg - 1) * T(0.5) * T(kPiR) / rx * fsin(arg);
    }
  }
  if (s.mode == 0 || s.mode == 3) {
    const T vb = pref * invp;
    const T dvb = -T(s.p) * vb * inv;
    if (s.mode == 0) {
      v = vb;
      if constexpr (DERIV) dv = dvb;
    } else {
      v = vb * (T(1) - fc);
      if constexpr (DERIV) dv = dvb * (T(1) - fc) - vb * dfc;
    }
    return;
  }
  const T x = dc * dc * T(s.inv_2s2);
  T Q, dens;
  upper_gamma<T>(s.p, x, Q, dens);
  const T dxdd = T(2) * dc * T(s.inv_2s2);
  if (s.mode == 1) {
    v = pref * Q * invp;
    if constexpr (DERIV) dv = pref * (-dens * dxdd * invp - T(s.p) * Q * invp * inv);
  } else {
    const T P = x < T(1) ? lower_gamma_series<T>(s.p, x) : T(1) - Q;
    const T vl = pref * P * invp;
    v = -vl * fc;
    if constexpr (DERIV) {
      const T dvl = pref * (dens * dxdd * invp - T(s.p) * P * invp * inv);
      dv = -(dvl * fc + vl * dfc);
    }
  }
}

// Fast path of the fused row kernels for range-separated potentials without exclusion (mode 1) and a compile-time
// exponent P:   v = pref Q(P/2, x) / d^P,   (dv/dd) / d = -pref (2 x dens + P Q) / d^(P+2),   x = d^2 / (2 sigma^2),
// dens = x^(P/2-1) e^-x / Gamma(P/2), evaluated from d^2.  The float version uses the hardware rsq / rcp / exp2 (1 ulp
// each) instead of IEEE division sequences and libm range reduction: the generic sr_eval costs ~190 VALU slots per pair,
// which made the fused kernels VALU-bound.  P = 1 is the Coulomb potential.
struct FastRS {
  double inv_2s2, c1, pref;  // c1 = 1/(sigma sqrt 2)
  double cheb[kErfcChebTerms];  // erfc coefficients of the fp64 path (see erfc_from_exp)
};
inline FastRS make_fast_rs(const SRPot& s) {
  FastRS f{s.inv_2s2, sqrt(s.inv_2s2), s.pref, MIPME_ERFC_CHEB};
  return f;
}
// exponents with a dedicated instantiation (others take the generic sr_eval): Coulomb and dispersion
inline int fast_rs_exponent(const SRPot& s) { return (s.mode == 1 && (s.p == 1 || s.p == 6)) ? s.p : 0; }

__device__ __forceinline__ float rs_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rs_rsqrt(double x) { return rsqrt_newton(x); }
__device__ __forceinline__ float rs_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rs_rcp(double x) { return rcp_newton(x); }
__device__ __forceinline__ float rs_exp_neg(float x) { return __builtin_amdgcn_exp2f(-1.4426950408889634f * x); }
__device__ __forceinline__ double rs_exp_neg(double x) { return exp_neg_fast(x); }
__device__ __forceinline__ float rs_erfc(float y, float e, const double*) { return erfc_from_exp_fast(y, e); }
__device__ __forceinline__ double rs_erfc(double y, double e, const double* c) { return erfc_from_exp(y, e, c); }

template <int P, bool DERIV, typename T>
__device__ __forceinline__ void fast_rs_eval(T inv_2s2, T c1, T pref, T d2, T& v, T& dvd, const double* cheb) {
  d2 = d2 < T(1e-30) ? T(1e-30) : d2;  // (NaN stays NaN)
  const T inv = rs_rsqrt(d2);
  const T inv2 = inv * inv;
  const T x = d2 * inv_2s2;
  const T e = rs_exp_neg(x);
  T Q, two_x_dens;  // Q(P/2, x) and 2 x dens(x)
  if constexpr (P % 2 == 0) {
    T term = T(1), sum = T(1);
#pragma unroll
    for (int k = 1; k < P / 2; ++k) {
      term *= x * T(1.0 / k);
      sum += term;
    }
    Q = e * sum;
    two_x_dens = T(2) * x * e * term;
  } else {
    // half-integer a = m + 1/2: term_k = x^(k-1/2) e^-x / Gamma(k+1/2); term_1 = 2 y e / sqrt(pi) with y = sqrt(x), so no
    // division by y is needed: Q = erfc(y) + sum_{k=1..m} term_k and 2 x dens = 2 x term_m = (2m+1) term_{m+1}
    constexpr int m = (P - 1) / 2;
    const T y = c1 * (d2 * inv);
    Q = rs_erfc(y, e, cheb);
    T term = T(2.0 * 0.56418958354775628695) * y * e;  // term_1
#pragma unroll
    for (int k = 1; k <= m; ++k) {
      Q += term;
      term *= x * T(1.0 / (k + 0.5));  // -> term_{k+1}
    }
    two_x_dens = T(2 * m + 1) * term;
  }
  T invp = inv;
#pragma unroll
  for (int k = 1; k < P; ++k) invp *= inv;
  const T pi = pref * invp;
  v = pi * Q;
  if constexpr (DERIV) dvd = -(pi * inv2) * (two_x_dens + T(P) * Q);
}

}  // namespace mipme
