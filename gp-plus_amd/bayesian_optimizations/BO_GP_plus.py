"""Cost-aware multi-fidelity Bayesian optimisation loop of the reference (``bayesian_optimizations/BO_GP_plus.py:11-216``)
on this build's GP_Plus / fit_model_scipy.

Per iteration: refit a GP_Plus on all data (every likelihood evaluation on the MI355X), then either minimise the
acquisition functions per fidelity with 12 random L-BFGS-B starts (``data_gen_func`` callable, :60-124) or score a pool
of candidates in one batched prediction per fidelity (``data_gen_func`` an array, :143-207), add the winner, pay its
cost, stop on the budget or when the incumbent stops moving.

Deviations, all forced by the back end: the 12 acquisition starts run sequentially (the reference ships the model to
joblib workers with dill; a GPU model does not pickle) and the model is built with ``qual_dict=qual_index`` on the GPU
(the reference passes ``qual_index`` positionally and an ``IS`` keyword its own constructor no longer has).
"""
from __future__ import annotations

import numpy as np
import torch
from scipy.optimize import minimize

from ..models import GP_Plus
from ..optim import fit_model_scipy
from .AFs import AF_HF, AF_LF, AF_HF_Engineering, AF_LF_Engineering

__all__ = ["BO"]


def BO(Xtrain=None, ytrain=None, costs=None, l_bound=None, u_bound=None, xmean=None, xstd=None, qual_index=None,
       data_gen_func=None, n_train=None, maximize_flag=False, one_iter=False, max_cost=40000, MF=True, AF_hf=AF_HF,
       AF_lf=AF_LF, max_iter=2, IS=True, device='cuda', fit_options=None, n_starts=12):
    ymin_list, xmin_list, cumulative_cost, bestf, Fidelity = [], [], [], [], []
    num_fidelity = list(qual_index.values())[-1]
    fit_options = {} if fit_options is None else fit_options

    def cost_fun(x):
        return costs[str(int(x))]

    def bestf_calculator(MF, num_fidelity, ytrain, Xtrain):
        pick = (lambda t: t.max()) if maximize_flag else (lambda t: t.min())
        if MF:
            return [pick(ytrain[Xtrain[:, -1] == i]).reshape(-1,).item() for i in range(num_fidelity)]
        return [pick(ytrain).reshape(-1,).item()]

    def fit(Xtrain, ytrain):
        model = GP_Plus(Xtrain, ytrain, qual_dict=qual_index, dtype=torch.float64, device=device)
        fit_model_scipy(model, bounds=True, options=fit_options)
        return model

    def run_scipy(EI, best_f, bound, model, xmean, xstd, fidelity):
        random_seed = np.random.choice(range(0, 1000), size=n_starts, replace=False)
        best = (np.inf, None)
        for k in range(n_starts):  # (the reference's joblib fan-out, :61-77, sequential here)
            np.random.seed(random_seed[k])
            x0 = np.random.uniform(list(l_bound) + [fidelity], list(u_bound) + [fidelity])
            x0[-1] = np.round(x0[-1])
            res = minimize(EI, x0.reshape(-1,), args=(best_f, model, xmean, xstd, cost_fun), bounds=bound)
            if res.fun < best[0]:
                best = (res.fun, res.x)
        return best

    if callable(data_gen_func):
        Xtrain = torch.as_tensor(Xtrain, dtype=torch.float64)
        ytrain = torch.as_tensor(ytrain, dtype=torch.float64).reshape(-1)
        initial_cost = np.sum([cost_fun(v) for v in Xtrain[:, -1]])
        cumulative_cost.append(initial_cost)
        problem = lambda x: data_gen_func(False, x)
        while cumulative_cost[-1] < max_cost:
            best_values = bestf_calculator(MF, num_fidelity, ytrain, Xtrain)
            bestf.append(best_values[0])
            if len(bestf) > max_iter and np.var(bestf[-max_iter:]) < 1e-6:
                break
            model = fit(Xtrain, ytrain)
            X_list, y_list = [], []
            for i in range(num_fidelity):
                bound = tuple(list(zip(l_bound, u_bound)) + [(i, i)])
                Y, X = run_scipy(AF_hf if i == 0 else AF_lf, best_values[0], bound, model, np.array(xmean), np.array(xstd), i)
                X_list.append(X)
                y_list.append(Y)
            temp = np.asarray(X_list[int(np.argmin(y_list))], dtype=np.float64)
            ynew = torch.as_tensor(problem(torch.tensor(temp).unsqueeze(0)), dtype=torch.float64)
            if MF:
                Xnew = np.concatenate([((temp[0:-1] - xmean) / xstd).reshape(1, -1), temp[-1].reshape(-1, 1)], axis=-1)
            else:
                Xnew = ((temp - xmean) / xstd).reshape(1, -1)
            Xtrain = torch.cat([Xtrain, torch.tensor(Xnew.reshape(1, -1))])
            ytrain = torch.cat([ytrain, ynew.reshape(-1,)])
            ymin_list.append(ynew.reshape(-1,))
            xmin_list.append(Xnew)
            cumulative_cost.append(initial_cost + cost_fun(Xnew[0][-1]))
            initial_cost = cumulative_cost[-1]
            Fidelity.append(Xnew[0][-1])
            if one_iter:
                bestf.append(bestf_calculator(MF, num_fidelity, ytrain, Xtrain)[0])
                break
    else:
        pool = np.asarray(data_gen_func, dtype=np.float64)  # columns: inputs ..., fidelity, response
        Xtrain = np.empty((0, pool.shape[1] - 1))
        ytrain = np.empty((0,))
        for i in range(num_fidelity):
            rows = pool[pool[:, -2] == i]
            idx = np.random.randint(0, len(rows), n_train[i])
            Xtrain = np.append(Xtrain, rows[idx][:, 0:-1], axis=0)
            ytrain = np.append(ytrain, rows[idx][:, -1], axis=0)  # (the reference indexes the whole pool here, :135)
        Xtrain, ytrain = torch.tensor(Xtrain), torch.tensor(ytrain)
        initial_cost = np.sum([cost_fun(v) for v in Xtrain[:, -1]])
        cumulative_cost.append(initial_cost)
        while cumulative_cost[-1] < max_cost:
            best_values = bestf_calculator(MF, num_fidelity, ytrain, Xtrain)
            bestf.append(best_values[0])
            if len(bestf) > max_iter and np.var(bestf[-max_iter:]) < 1e-6:
                break
            model = fit(Xtrain, ytrain)
            scores, owners = [], []
            for i in range(num_fidelity):
                sel = np.nonzero(pool[:, -2] == i)[0]
                cand = torch.tensor(pool[sel][:, 0:-1])
                with torch.no_grad():
                    ytest, ystd = model.predict(cand, return_std=True, include_noise=False)
                af = AF_HF_Engineering if i == 0 else AF_LF_Engineering
                scores.append(af(best_values[i], ytest.reshape(-1, 1), ystd.reshape(-1, 1), cand, cost_fun, maximize=maximize_flag))
                owners.append(sel)
            index = int(np.concatenate(owners)[int(torch.argmax(torch.cat(scores, dim=0)))])
            Xnew = torch.tensor(pool[index][0:-1])
            ynew = pool[index][-1]
            Xtrain = torch.cat([Xtrain, Xnew.reshape(1, -1)])
            ytrain = torch.cat([ytrain, torch.tensor(ynew).reshape(-1,)], dim=0)
            ymin_list.append(np.asarray(ynew).reshape(-1,))
            xmin_list.append(Xnew)
            cumulative_cost.append(initial_cost + cost_fun(Xnew[-1]))
            initial_cost = cumulative_cost[-1]
            Fidelity.append(Xnew[-1])
            if one_iter:
                bestf.append(bestf_calculator(MF, num_fidelity, ytrain, Xtrain)[0])
                break
    return np.array(bestf), np.array(cumulative_cost)
