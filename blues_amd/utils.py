"""Host logic of the switching path that lives in blues/utils.py in the reference."""
import logging
import sys
from math import ceil, floor

logger = logging.getLogger(__name__)


def calculateNCMCSteps(nstepsNC=0, nprop=1, propLambda=0.3, **kwargs):
    """Same contract as reference blues/utils.py:89-145: forces an even nstepsNC, derives the
    number of lambda-switching steps compatible with (nprop, propLambda) and returns
    {'nstepsNC','propSteps','moveStep','nprop','propLambda'}; exits for nstepsNC=1 as the reference does."""
    if (nstepsNC % 2) != 0:
        rounded_val = nstepsNC & ~1
        msg = 'nstepsNC=%i must be even for symmetric protocol.' % (nstepsNC)
        if rounded_val:
            logger.warning(msg + ' Setting to nstepsNC=%i' % rounded_val)
            nstepsNC = rounded_val
        else:
            logger.error(msg)
            sys.exit(1)
    lambdaSteps = nstepsNC / (2 * (nprop * propLambda + 0.5 - propLambda))
    lambdaSteps = int(lambdaSteps) if int(lambdaSteps) % 2 == 0 else int(lambdaSteps) + 1
    in_prop = int(nprop * (2 * floor(propLambda * lambdaSteps)))
    out_prop = int(2 * ceil((0.5 - propLambda) * lambdaSteps))
    propSteps = int(in_prop + out_prop)
    if propSteps != nstepsNC:
        logger.warning("nstepsNC=%s is incompatible with prop_lambda=%s and nprop=%s." % (nstepsNC, propLambda, nprop))
        logger.warning("Changing NCMC protocol to %s lambda switching within %s total propagation steps." % (lambdaSteps, propSteps))
        nstepsNC = lambdaSteps
    return {'nstepsNC': nstepsNC, 'propSteps': propSteps, 'moveStep': int(nstepsNC / 2), 'nprop': nprop, 'propLambda': propLambda}
